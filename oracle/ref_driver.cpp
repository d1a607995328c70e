/*
 * ref_driver.cpp - TEST INFRASTRUCTURE (oracle side), not product code.
 *
 * Headless driver for the UNMODIFIED reference engine, compiled by
 * oracle/Makefile against the sources where they lie under /root/reference
 * (nothing of the reference is copied into this repository).  It uses the
 * reference's public API only -- rt_Platform / rt_Scene::render / get_frame
 * (core/engine/engine.h:131-136, 283-330), the way test/core_test.cpp:184,
 * 939-1055 does -- plus one private runtime field, rt_Scene::depth
 * (engine.h:268, copied to s_inf->depth in engine.cpp:3608), to realise the
 * BASELINE.json configs that name a recursion depth (limit 1).
 *
 * What it produces:
 *   --out F.raw      the reference's frame, frm_w*frm_h little-endian uint32
 *   (stdout)         "hash <fnv1a64>" of the frame (pixel & 0xFFFFFF, row-major)
 *   --snapshot F.qrs the flattened scene (include/qr_scene.h) captured through
 *                    the drop-in shim (oracle/ref_shim.cpp -> qr_capture_snapshot);
 *                    only in the binary linked with the shim (qr_ref_shim)
 *   --animate MS     advance the scene time by MS per --bench frame (objects, lights and camera move, the
 *                    engine rebuilds its lists: nothing can be reused from the previous frame)
 *   --pt N           path-tracer mode: render N frames, the output is their running mean
 *   --bench N        wall-clock of N rt_Scene::render() calls (CPU baseline,
 *                    bench.py's cpu_baseline.kind == "reference")
 */
#define private public          /* rt_Scene::depth only, see header comment */
#define protected public
#include "engine.h"
#undef private
#undef protected

#include "all_scn.h"

#include "scn_test01.h"
#include "scn_test02.h"
#include "scn_test03.h"
#include "scn_test04.h"
#include "scn_test05.h"
#include "scn_test06.h"
#include "scn_test07.h"
#include "scn_test08.h"
#include "scn_test09.h"
#include "scn_test10.h"
#include "scn_test11.h"
#include "scn_test12.h"
#include "scn_test13.h"
#include "scn_test14.h"
#include "scn_test15.h"
#include "scn_test16.h"
#include "scn_test17.h"
#include "scn_test18.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#include <sys/mman.h>
#include <sys/time.h>
#include <pthread.h>
#include <vector>
#include <algorithm>

/* set by the shim-linked binary: where the next shim call writes its snapshot */
extern "C" { const char *qr_shim_snapshot_path = NULL; int qr_shim_calls = 0; int qr_shim_status = 0; double qr_shim_ms = 0.0; }

static double now_ms()
{
    timeval tm;
    gettimeofday(&tm, NULL);
    return tm.tv_sec * 1000.0 + tm.tv_usec / 1000.0;
}

static rt_pntr sys_alloc(rt_size size)
{
    rt_pntr ptr = mmap(NULL, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (ptr == MAP_FAILED || ptr == RT_NULL)
    {
        throw rt_Exception("alloc failed in ref_driver sys_alloc");
    }
    return ptr;
}

static rt_void sys_free(rt_pntr ptr, rt_size size)
{
    munmap(ptr, size);
}

/* ------------------------------------------------------------------------ */
/* optional thread pool: same contract as root/RooT_linux.cpp:546-793       */
/* (f_init/f_term/f_update/f_render hooks, engine.h:71-74), own design      */
/* ------------------------------------------------------------------------ */

struct Pool
{
    rt_Platform *pfm;
    int thnum;
    std::vector<pthread_t> th;
    pthread_barrier_t go, done;
    volatile int cmd;       /* 0 exit, 1 update, 2 render */
    volatile int phase;
};

struct PoolArg { Pool *pool; int index; };

static void *pool_worker(void *p)
{
    PoolArg *pa = (PoolArg *)p;
    Pool *pool = pa->pool;
    int index = pa->index;
    for (;;)
    {
        pthread_barrier_wait(&pool->go);
        int cmd = pool->cmd, phase = pool->phase;
        if (cmd == 0) break;
        rt_Scene *scn = pool->pfm->get_cur_scene();
        try
        {
            if (cmd == 1) scn->update_slice(index, phase);
            if (cmd == 2) scn->render_slice(index, phase);
        }
        catch (rt_Exception e)
        {
            fprintf(stderr, "worker %d exception: %s\n", index, e.err);
        }
        pthread_barrier_wait(&pool->done);
    }
    delete pa;
    return NULL;
}

static int g_threads = 1;

static rt_pntr pool_init(rt_si32 thnum, rt_Platform *pfm)
{
    Pool *pool = new Pool;
    pool->pfm = pfm;
    pool->thnum = g_threads;
    pfm->set_thnum(g_threads);
    pthread_barrier_init(&pool->go, NULL, g_threads + 1);
    pthread_barrier_init(&pool->done, NULL, g_threads + 1);
    pool->th.resize(g_threads);
    for (int i = 0; i < g_threads; i++)
    {
        PoolArg *pa = new PoolArg; pa->pool = pool; pa->index = i;
        pthread_create(&pool->th[i], NULL, pool_worker, pa);
    }
    return pool;
}

static rt_void pool_term(rt_pntr tdata, rt_si32 thnum)
{
    Pool *pool = (Pool *)tdata;
    pool->cmd = 0;
    pthread_barrier_wait(&pool->go);
    for (size_t i = 0; i < pool->th.size(); i++) pthread_join(pool->th[i], NULL);
    delete pool;
}

static rt_void pool_update(rt_pntr tdata, rt_si32 thnum, rt_si32 phase)
{
    Pool *pool = (Pool *)tdata;
    pool->cmd = 1; pool->phase = phase;
    pthread_barrier_wait(&pool->go);
    pthread_barrier_wait(&pool->done);
}

static rt_void pool_render(rt_pntr tdata, rt_si32 thnum, rt_si32 phase)
{
    Pool *pool = (Pool *)tdata;
    pool->cmd = 2; pool->phase = phase;
    pthread_barrier_wait(&pool->go);
    pthread_barrier_wait(&pool->done);
}

/* ------------------------------------------------------------------------ */

static rt_SCENE *find_scene(const char *name)
{
    if (!strcmp(name, "demo01")) return &scn_demo01::sc_root;
    if (!strcmp(name, "demo02")) return &scn_demo02::sc_root;
    if (!strcmp(name, "demo03")) return &scn_demo03::sc_root;
    if (!strcmp(name, "test01")) return &scn_test01::sc_root;
    if (!strcmp(name, "test02")) return &scn_test02::sc_root;
    if (!strcmp(name, "test03")) return &scn_test03::sc_root;
    if (!strcmp(name, "test04")) return &scn_test04::sc_root;
    if (!strcmp(name, "test05")) return &scn_test05::sc_root;
    if (!strcmp(name, "test06")) return &scn_test06::sc_root;
    if (!strcmp(name, "test07")) return &scn_test07::sc_root;
    if (!strcmp(name, "test08")) return &scn_test08::sc_root;
    if (!strcmp(name, "test09")) return &scn_test09::sc_root;
    if (!strcmp(name, "test10")) return &scn_test10::sc_root;
    if (!strcmp(name, "test11")) return &scn_test11::sc_root;
    if (!strcmp(name, "test12")) return &scn_test12::sc_root;
    if (!strcmp(name, "test13")) return &scn_test13::sc_root;
    if (!strcmp(name, "test14")) return &scn_test14::sc_root;
    if (!strcmp(name, "test15")) return &scn_test15::sc_root;
    if (!strcmp(name, "test16")) return &scn_test16::sc_root;
    if (!strcmp(name, "test17")) return &scn_test17::sc_root;
    if (!strcmp(name, "test18")) return &scn_test18::sc_root;
    return NULL;
}

static uint64_t fnv1a64_frame(const rt_ui32 *frame, int w, int h, int row)
{
    uint64_t hsh = 0xcbf29ce484222325ull;
    for (int y = 0; y < h; y++)
    {
        const rt_ui32 *p = frame + (ptrdiff_t)y * row;
        for (int x = 0; x < w; x++)
        {
            uint32_t v = p[x] & 0x00FFFFFF;
            for (int b = 0; b < 4; b++)
            {
                hsh ^= (v >> (8 * b)) & 0xFF;
                hsh *= 0x100000001b3ull;
            }
        }
    }
    return hsh;
}

static void usage()
{
    fprintf(stderr,
        "usage: qr_ref --scene NAME [-w W] [-h H] [-t MS] [--fsaa 0|2|4] [--gamma] [--fresnel]\n"
        "              [--depth D] [--simd N,K,S] [--opts none|full] [--threads T]\n"
        "              [--out F.raw] [--snapshot F.qrs] [--tree F.json] [--bench N] [--camera K] [--pt N] [--pt-warm] [--opts-off tiling,varray,..] [--shim] [--jitter SEED] [--swarm N,SEED[,MIX]]\n");
}


/*
 * --tree F.json: the object hierarchy as the engine holds it after the frame at time T -- per object the inputs of
 * the hierarchical update (parent, tag, rt_TRANSFORM3D after the animators ran, shape parameters) and its results
 * (matrix, trnode, transform flags), plus the snapshot index (qr_capture_index) of the SIMD records each object owns.
 * Fixture for include/qr_hierarchy.h (tests/test_hierarchy.py); floats are written as their 32 bits in hex.
 * The engine keeps these members private; this TU is compiled with -fno-access-control (oracle/Makefile).
 */
extern "C" int qr_capture_index(int kind, const void *record) __attribute__((weak));   /* libqrhip: qr_ref_shim only */
extern "C" int qr_frame_register(void *frame, unsigned long long bytes) __attribute__((weak));
extern "C" int qr_frame_unregister(void *frame) __attribute__((weak));
static int capture_index(int kind, const void *r) { return qr_capture_index ? qr_capture_index(kind, r) : -1; }

static void put_f(FILE *f, const char *key, const rt_real *v, int n)
{
    fprintf(f, "\"%s\": [", key);
    for (int i = 0; i < n; i++)
    {
        uint32_t u; float x = (float)v[i]; memcpy(&u, &x, 4);
        fprintf(f, "%s\"%08x\"", i ? ", " : "", u);
    }
    fprintf(f, "]");
}

static void dump_node(FILE *f, rt_Object *o, int parent, std::vector<rt_Object *> &seen, bool &first)
{
    const int me = (int)seen.size();
    seen.push_back(o);
    fprintf(f, "%s\n  {\"parent\": %d, \"tag\": %d, \"anim\": %d, ", first ? "" : ",", parent, (int)o->tag, o->obj->f_anim != RT_NULL ? 1 : 0);
    first = false;
    put_f(f, "scl", o->trm->scl, 3); fprintf(f, ", ");
    put_f(f, "rot", o->trm->rot, 3); fprintf(f, ", ");
    put_f(f, "pos", o->trm->pos, 3); fprintf(f, ", ");
    rt_real shape[3] = { 0, 0, 0 };
    if (o->tag > RT_TAG_PLANE && o->tag < RT_TAG_SURFACE_MAX)
    {
        const rt_real *p = (const rt_real *)((rt_SURFACE *)o->obj->obj.pobj + 1);   /* parameters follow rt_SURFACE, format.h:496-727 */
        const int n = (o->tag == RT_TAG_HYPERBOLOID || o->tag == RT_TAG_HYPERCYLINDER || o->tag == RT_TAG_HYPERPARABOLOID) ? 2 : 1;
        for (int i = 0; i < n; i++) shape[i] = p[i];
    }
    put_f(f, "shape", shape, 3); fprintf(f, ", ");
    put_f(f, "mtx", &o->mtx[0][0], 16); fprintf(f, ", ");
    int trn = -1;
    for (size_t i = 0; i < seen.size(); i++) if (seen[i] == o->trnode) trn = (int)i;
    fprintf(f, "\"trnode\": %d, \"obj_has_trm\": %d, \"mtx_has_trm\": %d", trn, (int)o->obj_has_trm, (int)o->mtx_has_trm);
    if (o->tag == RT_TAG_LIGHT)
    {
        fprintf(f, ", \"lgt\": %d", capture_index(1, ((rt_Light *)o)->s_lgt));
    }
    else if (o->tag == RT_TAG_CAMERA)
    {
        fprintf(f, ", "); put_f(f, "pov", &((rt_Camera *)o)->pov, 1);
    }
    else
    {
        rt_Node *nd = (rt_Node *)o;
        fprintf(f, ", \"srf\": %d", capture_index(0, nd->s_srf));
        int bvn = -1;
        for (size_t i = 0; i < seen.size(); i++) if (seen[i] == nd->bvnode) bvn = (int)i;
        fprintf(f, ", \"bvnode\": %d", bvn);
    }
    if (o->tag >= RT_TAG_PLANE && o->tag < RT_TAG_SURFACE_MAX)
    {
        /* bounds (rt_Surface::update_minmax / update_bounds, object.cpp:2690-2845): the inputs -- the axis clippers of the
         * scene description -- and the engine's results */
        rt_Surface *sf = (rt_Surface *)o;
        const rt_SURFACE *sd = (const rt_SURFACE *)o->obj->obj.pobj;
        fprintf(f, ", "); put_f(f, "lmin", sd->min, 3); fprintf(f, ", "); put_f(f, "lmax", sd->max, 3);
        fprintf(f, ", "); put_f(f, "bmin", sf->shape->bmin, 3); fprintf(f, ", "); put_f(f, "bmax", sf->shape->bmax, 3);
        fprintf(f, ", "); put_f(f, "cmin", sf->shape->cmin, 3); fprintf(f, ", "); put_f(f, "cmax", sf->shape->cmax, 3);
        fprintf(f, ", "); put_f(f, "mid", sf->bvbox->mid, 3); fprintf(f, ", "); put_f(f, "rad", &sf->bvbox->rad, 1);
        fprintf(f, ", \"verts_num\": %d", (int)sf->bvbox->verts_num);
        if (o->tag == RT_TAG_PLANE)
        {
            /* texture scale and position of the two materials before the plane's axis scalers enter them
             * (rt_Plane::update_fields, object.cpp:2893-2938): outer scl x, y, pos x, y, then inner */
            const rt_real tex[8] = { sf->outer->scl[0], sf->outer->scl[1], sf->outer->sd->pos[0], sf->outer->sd->pos[1],
                                     sf->inner->scl[0], sf->inner->scl[1], sf->inner->sd->pos[0], sf->inner->sd->pos[1] };
            fprintf(f, ", "); put_f(f, "tex", tex, 8);
        }
    }
    if (o->tag == RT_TAG_ARRAY)
    {
        rt_Array *a = (rt_Array *)o;
        rt_BOUND *bx[3] = { a->inbox, a->bvbox, a->trbox };
        static const char *bn[3] = { "inbox", "bvbox", "trbox" };
        for (int b = 0; b < 3; b++)
        {
            char key[32];
            snprintf(key, sizeof(key), "%s_min", bn[b]); fprintf(f, ", "); put_f(f, key, bx[b]->bmin, 3);
            snprintf(key, sizeof(key), "%s_max", bn[b]); fprintf(f, ", "); put_f(f, key, bx[b]->bmax, 3);
            snprintf(key, sizeof(key), "%s_mid", bn[b]); fprintf(f, ", "); put_f(f, key, bx[b]->mid, 3);
            snprintf(key, sizeof(key), "%s_rad", bn[b]); fprintf(f, ", "); put_f(f, key, &bx[b]->rad, 1);
        }
        fprintf(f, ", \"inb\": %d, \"bvb\": %d}", capture_index(0, a->s_inb), capture_index(0, a->s_bvb));
        for (int i = 0; i < a->obj_num; i++) dump_node(f, a->obj_arr[i], me, seen, first);
    }
    else
    {
        fprintf(f, "}");
    }
}

static bool dump_tree(rt_Scene *sc, const char *scene, long time_ms, const char *path)
{
    FILE *f = fopen(path, "w");
    if (f == NULL) { fprintf(stderr, "cannot open %s\n", path); return false; }
    std::vector<rt_Object *> seen;
    fprintf(f, "{\"scene\": \"%s\", \"time\": %ld, \"opts\": %d, \"x_res\": %d, \"y_res\": %d, \"nodes\": [", scene, time_ms,
            (int)sc->opts, (int)sc->x_res, (int)sc->y_res);
    bool first = true;
    dump_node(f, sc->root, -1, seen, first);
    int cam = -1;
    for (size_t i = 0; i < seen.size(); i++) if (seen[i] == (rt_Object *)sc->cam) cam = (int)i;
    fprintf(f, "\n], \"camera\": %d}\n", cam);
    fclose(f);
    return true;
}

/*
 * --jitter SEED: before the engine sees the scene, every object of its static description gets another transform --
 * right-angle and arbitrary rotations, unit, negative and non-unit scalers, shifted positions -- so that the hierarchical
 * update meets combinations the stock scenes do not have (fixtures tests/golden/tree/fuzz_*; the frame is not kept).
 */
static uint64_t g_jit = 0;
static int g_shift_only = 0;        /* --shift SEED: positions only (nothing else of a transform changes: a scene "a moment later") */
static uint32_t jit_next() { g_jit ^= g_jit << 13; g_jit ^= g_jit >> 7; g_jit ^= g_jit << 17; return (uint32_t)(g_jit >> 11); }
static void jitter_tree(rt_OBJECT *arr, int n, int level)
{
    static const rt_real right[] = { -270.0f, -180.0f, -90.0f, 0.0f, 90.0f, 180.0f, 270.0f };
    static const rt_real scales[] = { 0.5f, 2.0f, 1.5f, -2.0f, -1.0f, 1.25f };
    for (int i = 0; i < n; i++)
    {
        rt_OBJECT *o = &arr[i];
        const bool cam = o->obj.tag == RT_TAG_CAMERA;
        if (g_shift_only)
        {
            /* two thirds of the objects move by up to +-0.5 along every axis; the camera stays */
            if (!cam && jit_next() % 3 != 0)
                for (int a = 0; a < 3; a++) o->trm.pos[a] += (rt_real)((int)(jit_next() % 101) - 50) / 100.0f;
            if (o->obj.tag == RT_TAG_ARRAY && level < 16) jitter_tree((rt_OBJECT *)o->obj.pobj, o->obj.obj_num, level + 1);
            continue;
        }
        const uint32_t mode = jit_next() % 10;
        for (int a = 0; a < 3 && !cam; a++)
        {
            if (mode < 5) o->trm.rot[a] = right[jit_next() % 7];                               /* stays trivial */
            else if (mode < 8) o->trm.rot[a] = (rt_real)((int)(jit_next() % 3600) - 1800) / 10.0f;
            /* else: as the scene has it */
        }
        const uint32_t sm = jit_next() % 10;
        for (int a = 0; a < 3 && !cam; a++)
        {
            if (sm == 0) o->trm.scl[a] = -1.0f;
            else if (sm < 3) o->trm.scl[a] = scales[jit_next() % 6];
            else if (sm < 5) o->trm.scl[a] = 1.0f;
        }
        for (int a = 0; a < 3; a++) o->trm.pos[a] += (rt_real)((int)(jit_next() % 401) - 200) / 100.0f;
        if (o->obj.tag == RT_TAG_ARRAY && level < 16) jitter_tree((rt_OBJECT *)o->obj.pobj, o->obj.obj_num, level + 1);
    }
}

/*
 * --swarm N,SEED: N more spheres in the scene, in arrays of up to 12 under one new array of the root -- every group with
 * a bounding-volume relation, every fourth group rotated (a transform node: its spheres live in its space), some spheres
 * clipped, materials from plain to glass (none textured: the reference computes no texture coordinates for quadrics --
 * QD_mat, tracer.cpp:4845-4905, has no texture block -- and reads whatever the context still holds from the last plane).  The engine builds its own lists for them; the fixture pins the kernel and the
 * list-building pass on a crowd of small quadrics against the reference itself (the synthetic 10 000-object scene of
 * BASELINE config 5 is of this kind, but the engine's per-surface lists grow with N^2: a few hundred is what fits).
 */
/* the largest of the quadric records (format.h:496-727): rt_SURFACE + two parameters */
struct SwarmShape { rt_SURFACE srf; rt_real p0, p1; };

static void add_swarm(rt_SCENE *scn, int n, uint64_t seed, int mix)
{
    static rt_MATERIAL *outer[] = { &mt_plain01_red01, &mt_plain01_blue01, &mt_metal01_cyan01, &mt_metal02_orange01,
                                    &mt_metal03_nickel01, &mt_glass01_orange01, &mt_air_to_glass03, &mt_plain01_cyan01,
                                    &mt_plain01_green01, &mt_metal01_pink01 };
    g_jit = 0xD1B54A32D192ED03ull * (seed + 1);
    const int per = 12, n_groups = (n + per - 1) / per;
    SwarmShape *sph = (SwarmShape *)calloc((size_t)n, sizeof(SwarmShape));
    rt_OBJECT *objs = (rt_OBJECT *)calloc((size_t)n, sizeof(rt_OBJECT));
    rt_OBJECT *groups = (rt_OBJECT *)calloc((size_t)n_groups, sizeof(rt_OBJECT));
    rt_RELATION *rels = (rt_RELATION *)calloc((size_t)n_groups, sizeof(rt_RELATION));
    auto unit = [](rt_TRANSFORM3D &t) { for (int a = 0; a < 3; a++) { t.scl[a] = 1.0f; t.rot[a] = 0.0f; t.pos[a] = 0.0f; } };
    for (int i = 0; i < n; i++)
    {
        SwarmShape &sp = sph[i];
        for (int a = 0; a < 3; a++) { sp.srf.min[a] = -RT_INF; sp.srf.max[a] = +RT_INF; }
        const rt_real rad = 0.15f + (rt_real)(jit_next() % 46) / 100.0f;
        sp.p0 = rad;
        int tag = RT_TAG_SPHERE;
        if (jit_next() % 5 == 0 && !getenv("QR_SWARM_NOBOWL")) sp.srf.max[RT_K] = rad * 0.6f;                 /* an open bowl */
        if (mix)
        {
            /* other quadrics, all cut to a finite piece along their axis (MIX): cylinder rad; cone rat; paraboloid par;
             * hyperboloid rat, hyp (format.h:496-655) */
            switch (jit_next() % 6)
            {
            case 0: tag = RT_TAG_CYLINDER; sp.p0 = rad * 0.6f; sp.srf.min[RT_K] = -rad; sp.srf.max[RT_K] = rad; break;
            case 1: tag = RT_TAG_CONE; sp.p0 = 0.5f + (rt_real)(jit_next() % 11) / 10.0f; sp.srf.min[RT_K] = -rad; sp.srf.max[RT_K] = (jit_next() & 1) ? 0.0f : rad; break;
            case 2: tag = RT_TAG_PARABOLOID; sp.p0 = 0.3f + (rt_real)(jit_next() % 8) / 10.0f; sp.srf.min[RT_K] = -RT_INF; sp.srf.max[RT_K] = rad; break;
            case 3: tag = RT_TAG_HYPERBOLOID; sp.p0 = 0.6f + (rt_real)(jit_next() % 9) / 10.0f; sp.p1 = rad * rad * 0.1f; sp.srf.min[RT_K] = -rad; sp.srf.max[RT_K] = rad; break;
            default: break;                     /* stays a sphere */
            }
        }
        rt_MATERIAL *m = outer[jit_next() % 10];
        rt_SIDE side = { { 1.0f, 1.0f }, 0.0f, { 0.0f, 0.0f }, m };
        sp.srf.side_outer = side;
        side.pmat = m == &mt_air_to_glass03 ? &mt_glass03_to_air : &mt_plain01_gray01;
        sp.srf.side_inner = side;
        rt_OBJECT &o = objs[i];
        unit(o.trm);
        o.trm.pos[RT_X] = (rt_real)((int)(jit_next() % 2401) - 1200) / 100.0f;
        o.trm.pos[RT_Y] = (rt_real)((int)(jit_next() % 2401) - 1200) / 100.0f;
        o.trm.pos[RT_Z] = 0.4f + (rt_real)(jit_next() % 701) / 100.0f;
        if (jit_next() % 7 == 0 && !getenv("QR_SWARM_NOSCALE")) o.trm.scl[RT_Z] = 1.5f;                           /* an ellipsoid */
        o.obj.tag = tag; o.obj.pobj = &sp; o.obj.obj_num = 1;
        if (mix && jit_next() % 3 == 0) { o.trm.rot[RT_X] = (rt_real)(90 * (int)(jit_next() % 4)); o.trm.rot[RT_Y] = (rt_real)(90 * (int)(jit_next() % 4)); }   /* axis maps */
        o.time = -1;
    }
    for (int g = 0; g < n_groups; g++)
    {
        rt_OBJECT &o = groups[g];
        unit(o.trm);
        if (g % 4 == 3 && !getenv("QR_SWARM_NOROT")) { o.trm.rot[RT_Z] = (rt_real)(jit_next() % 360); o.trm.rot[RT_X] = (rt_real)((int)(jit_next() % 41) - 20); }
        o.obj.tag = RT_TAG_ARRAY; o.obj.pobj = &objs[g * per]; o.obj.obj_num = (g + 1) * per <= n ? per : n - g * per;
        o.time = -1;
        rels[g].obj1 = -1; rels[g].rel = RT_REL_BOUND_ARRAY; rels[g].obj2 = g;
    }
    const int old_n = scn->root.obj_num;
    rt_OBJECT *kids = (rt_OBJECT *)calloc((size_t)old_n + 1, sizeof(rt_OBJECT));
    memcpy(kids, scn->root.pobj, (size_t)old_n * sizeof(rt_OBJECT));
    rt_OBJECT &sw = kids[old_n];
    unit(sw.trm);
    sw.obj.tag = RT_TAG_ARRAY; sw.obj.pobj = groups; sw.obj.obj_num = n_groups; sw.obj.prel = rels; sw.obj.rel_num = n_groups;
    sw.time = -1;
    scn->root.pobj = kids; scn->root.obj_num = old_n + 1;
}

int main(int argc, char **argv)
{
    int pt_frames = 0;
    const char *scene_name = NULL, *out_path = NULL, *snap_path = NULL, *opts_mode = NULL, *tree_path = NULL, *opts_off = NULL;
    int pt_warm = 0;
    int w = 640, h = 480, fsaa = 0, depth = -1, bench = 0, gamma = 0, fresnel = 0, camera = 0, gpu = 0, pin_frame = 0;
    int n_simd = 0, k_size = 0, s_type = 0;
    long time_ms = 0, animate_ms = 0, swarm_seed = 0;
    int swarm_n = 0, swarm_mix = 0;

    for (int i = 1; i < argc; i++)
    {
        if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene_name = argv[++i];
        else if (!strcmp(argv[i], "-w") && i + 1 < argc) w = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-h") && i + 1 < argc) h = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) time_ms = atol(argv[++i]);
        else if (!strcmp(argv[i], "--fsaa") && i + 1 < argc) fsaa = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gamma")) gamma = 1;
        else if (!strcmp(argv[i], "--fresnel")) fresnel = 1;
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--simd") && i + 1 < argc) sscanf(argv[++i], "%d,%d,%d", &n_simd, &k_size, &s_type);
        else if (!strcmp(argv[i], "--opts") && i + 1 < argc) opts_mode = argv[++i];
        else if (!strcmp(argv[i], "--opts-off") && i + 1 < argc) opts_off = argv[++i];
        else if (!strcmp(argv[i], "--pt-warm")) pt_warm = 1;
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) g_threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else if (!strcmp(argv[i], "--snapshot") && i + 1 < argc) snap_path = argv[++i];
        else if (!strcmp(argv[i], "--bench") && i + 1 < argc) bench = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpu")) gpu = 1;
        else if (!strcmp(argv[i], "--pin-frame")) pin_frame = 1;      /* what the binding does where the engine allocates its frame: qr_frame_register */
        else if (!strcmp(argv[i], "--animate") && i + 1 < argc) animate_ms = atol(argv[++i]);
        else if (!strcmp(argv[i], "--camera") && i + 1 < argc) camera = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--pt") && i + 1 < argc) pt_frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tree") && i + 1 < argc) tree_path = argv[++i];
        else if (!strcmp(argv[i], "--shim")) { n_simd = 1; s_type = 8; k_size = 1; }
        else if (!strcmp(argv[i], "--swarm") && i + 1 < argc) sscanf(argv[++i], "%d,%ld,%d", &swarm_n, &swarm_seed, &swarm_mix);
        else if (!strcmp(argv[i], "--shift") && i + 1 < argc) { g_jit = 0xC2B2AE3D27D4EB4Full * (uint64_t)(atol(argv[++i]) + 1); g_shift_only = 1; }
        else if (!strcmp(argv[i], "--jitter") && i + 1 < argc) g_jit = 0x9E3779B97F4A7C15ull * (uint64_t)(atol(argv[++i]) + 1);   /* every frame through ref_shim.cpp -> qr_render0 */
        else { usage(); return 2; }
    }
    if (scene_name == NULL) { usage(); return 2; }
    rt_SCENE *scn = find_scene(scene_name);
    if (scn == NULL) { fprintf(stderr, "unknown scene %s\n", scene_name); return 2; }
    if (g_threads < 1) g_threads = 1;
    if (swarm_n > 0 && scn->root.tag == RT_TAG_ARRAY) { const uint64_t keep = g_jit; add_swarm(scn, swarm_n, (uint64_t)swarm_seed, swarm_mix); g_jit = keep; }
    if (g_jit != 0 && scn->root.tag == RT_TAG_ARRAY) jitter_tree((rt_OBJECT *)scn->root.pobj, scn->root.obj_num, 0);

    int rc = 0;
    try
    {
        rt_Platform *pfm = g_threads > 1
            ? new rt_Platform(sys_alloc, sys_free, g_threads, pool_init, pool_term, pool_update, pool_render)
            : new rt_Platform(sys_alloc, sys_free, 1);

        /* Gamma / Fresnel: scene opts DISABLE optimisations, and these two
         * "optimisations" are the features being off (format.h:59-60, 73-75) */
        if (gamma)   scn->opts |= RT_OPTS_GAMMA;
        if (fresnel) scn->opts |= RT_OPTS_FRESNEL;

        int simd = pfm->set_simd(simd_init(n_simd, s_type, k_size));
        pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);

        int x_row = (w + RT_SIMD_WIDTH - 1) & ~(RT_SIMD_WIDTH - 1);

        rt_Scene *sc = new(pfm) rt_Scene(scn, w, h, x_row, RT_NULL, pfm);
        if (opts_mode != NULL)
        {
            sc->set_opts(!strcmp(opts_mode, "none") ? RT_OPTS_NONE : RT_OPTS_FULL);
        }
        if (opts_off != NULL)
        {
            /* --opts-off NAME[,NAME]: every optimisation on except the named ones (format.h:40-106); "tiling" is the
             * screen tiling of engine.cpp:1956-2128, 3129-3253 together with its margin extension */
            rt_si32 off = 0;
            if (strstr(opts_off, "tiling")) off |= RT_OPTS_TILING | RT_OPTS_TILING_EXT1;
            if (strstr(opts_off, "varray")) off |= RT_OPTS_VARRAY;
            if (strstr(opts_off, "tarray")) off |= RT_OPTS_TARRAY;
            if (strstr(opts_off, "2sided")) off |= RT_OPTS_2SIDED | RT_OPTS_2SIDED_EXT1 | RT_OPTS_2SIDED_EXT2;
            if (strstr(opts_off, "shadow")) off |= RT_OPTS_SHADOW | RT_OPTS_SHADOW_EXT1 | RT_OPTS_SHADOW_EXT2;
            if (strstr(opts_off, "render")) off |= RT_OPTS_RENDER;
            if (off == 0) { fprintf(stderr, "--opts-off: no known optimisation in '%s'\n", opts_off); return 2; }
            sc->set_opts(RT_OPTS_FULL & ~off);
        }
        if (depth >= 0)
        {
            if (depth > RT_STACK_DEPTH) depth = RT_STACK_DEPTH;
            sc->depth = depth;
        }
        for (int k = 0; k < camera; k++) sc->next_cam();

        /* --pt N: path-tracer mode (rt_Scene::set_pton, engine.cpp:3729): N frames accumulate into the engine's colour
         * planes, the frame shows their running mean */
        if (pt_frames > 0)
        {
            if (pt_warm || getenv("QR_REF_PT_WARM"))
            {
                /* experiment: one path-traced frame, then restart the accumulation (seeds and planes are reset) */
                sc->set_pton(1); sc->render(time_ms); sc->set_pton(0); sc->render(time_ms);   /* the ray-traced frame resets the sample count, tracer.cpp:1128-1132 */
            }
            sc->set_pton(1);
            for (int k = 1; k < pt_frames; k++) sc->render(time_ms);
        }
        sc->render(time_ms);
        rt_ui32 *frame = sc->get_frame();
        int row = sc->get_x_row();

        printf("scene %s w %d h %d row %d t %ld fsaa %d gamma %d fresnel %d depth %d simd %dx%dv%d threads %d\n",
               scene_name, w, h, row, time_ms, fsaa, gamma, fresnel, (int)sc->depth,
               (simd & 0xFF) * 128, (simd >> 16) & 0xFF, (simd >> 8) & 0xFF, pfm->get_thnum());
        const uint64_t cpu_hash = fnv1a64_frame(frame, w, h, row);      /* before --bench --animate moves the scene on */
        printf("hash %016llx\n", (unsigned long long)cpu_hash);

        if (out_path != NULL)
        {
            FILE *f = fopen(out_path, "wb");
            if (f == NULL) { fprintf(stderr, "cannot open %s\n", out_path); return 3; }
            for (int y = 0; y < h; y++) fwrite(frame + (ptrdiff_t)y * row, 4, w, f);
            fclose(f);
        }

        if (gpu)
        {
            /* DROP-IN TEST: same engine, same scene object, but the backend namespace occupied by
             * oracle/ref_shim.cpp -> qr_render0 -> HIP kernel.  The frame must equal the CPU SIMD frame. */
            int got = pfm->set_simd(simd_init(1, 8, 1));
            if (((got >> 8) & 0xFF) != 8 || (got & 0xFF) != 1)
            {
                fprintf(stderr, "shim target 128x1v8 not selectable (got %x); is this qr_ref_shim?\n", got);
                return 4;
            }
            pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);
            memset(frame, 0, (size_t)row * h * 4);
            if (pin_frame)
            {
                if (qr_frame_register == NULL || row < w) { fprintf(stderr, "--pin-frame: not available\n"); return 4; }
                const int prc = qr_frame_register(frame, (unsigned long long)row * h * 4);
                printf("pin_frame rc %d\n", prc);
                if (prc != 0) return 4;
            }
            qr_shim_snapshot_path = NULL;
            qr_shim_calls = 0;
            sc->render(time_ms);
            uint64_t gpu_hash = fnv1a64_frame(sc->get_frame(), w, h, sc->get_x_row());
            printf("gpu_hash %016llx shim_calls %d %s\n", (unsigned long long)gpu_hash, qr_shim_calls,
                   gpu_hash == cpu_hash ? "MATCH" : "MISMATCH");
            if (gpu_hash != cpu_hash) rc = 6;
            pfm->set_simd(simd_init(n_simd, s_type, k_size));           /* back to the engine's own backend */
            pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);
        }

        long t_next = time_ms;          /* the scene time only moves forward */
        if (bench > 0)
        {
            std::vector<double> ms;
            for (int i = 0; i < bench; i++)
            {
                double t0 = now_ms();
                sc->render(t_next);
                ms.push_back(now_ms() - t0);
                t_next += animate_ms;
            }
            std::sort(ms.begin(), ms.end());
            double sum = 0; for (double v : ms) sum += v;
            printf("bench frames %d animate_ms %ld min_ms %.3f median_ms %.3f mean_ms %.3f\n",
                   bench, animate_ms, ms[0], ms[ms.size() / 2], sum / ms.size());
        }

        if (gpu && bench > 0)
        {
            pfm->set_simd(simd_init(1, 8, 1));
            pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);
            std::vector<double> ms, in_call;
            long t_last = t_next;
            for (int i = 0; i < bench; i++)
            {
                qr_shim_ms = 0.0;
                double t0 = now_ms();
                sc->render(t_next);
                ms.push_back(now_ms() - t0);
                in_call.push_back(qr_shim_ms);
                t_last = t_next;
                t_next += animate_ms;
            }
            std::sort(ms.begin(), ms.end()); std::sort(in_call.begin(), in_call.end());
            /* median_ms: the whole rt_Scene::render (the engine's own update of the scene + the backend call);
             * call_median_ms: inside qr_render0 only (flatten + compile + upload + kernel + copy back) */
            printf("gpu_bench frames %d animate_ms %ld min_ms %.3f median_ms %.3f call_median_ms %.3f engine_side_median_ms %.3f (engine update | flatten + compile + upload + kernel + copy back)\n",
                   bench, animate_ms, ms[0], ms[ms.size() / 2], in_call[in_call.size() / 2], ms[ms.size() / 2] - in_call[in_call.size() / 2]);
            if (animate_ms != 0)
            {
                /* the last animated frame once more on both backends: the frames must be equal */
                sc->render(t_last);
                const uint64_t g = fnv1a64_frame(sc->get_frame(), w, h, sc->get_x_row());
                pfm->set_simd(simd_init(n_simd, s_type, k_size));
                pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);
                sc->render(t_last);
                const uint64_t c2 = fnv1a64_frame(sc->get_frame(), w, h, sc->get_x_row());
                printf("animated frame t %ld gpu_hash %016llx cpu_hash %016llx %s\n", t_last,
                       (unsigned long long)g, (unsigned long long)c2, g == c2 ? "ANIM_MATCH" : "ANIM_MISMATCH");
                if (g != c2) rc = 6;
            }
        }

        if (snap_path != NULL)
        {
            /* select the namespace the shim occupies: simd_128v8 (n=1,k=1,s=8) */
            int got = pfm->set_simd(simd_init(1, 8, 1));
            if (((got >> 8) & 0xFF) != 8 || (got & 0xFF) != 1)
            {
                fprintf(stderr, "shim target 128x1v8 not selectable (got %x); is this qr_ref_shim?\n", got);
                return 4;
            }
            pfm->set_fsaa(fsaa == 4 ? RT_FSAA_4X : fsaa == 2 ? RT_FSAA_2X : RT_FSAA_NO);
            qr_shim_snapshot_path = snap_path;
            qr_shim_calls = 0;
            sc->render(time_ms);
            qr_shim_snapshot_path = NULL;
            if (qr_shim_calls != pfm->get_thnum() || qr_shim_status != 0)
            {
                fprintf(stderr, "shim was called %d times, status %d\n", qr_shim_calls, qr_shim_status);
                return 5;
            }
            printf("snapshot %s\n", snap_path);
        }

        if (tree_path != NULL)
        {
            if (!dump_tree(sc, scene_name, time_ms, tree_path)) return 3;
            printf("tree %s\n", tree_path);
        }

        delete sc;
        delete pfm;
    }
    catch (rt_Exception e)
    {
        fprintf(stderr, "exception: %s\n", e.err);
        rc = 1;
    }
    return rc;
}
