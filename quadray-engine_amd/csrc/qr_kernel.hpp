/*
 * qr_kernel.hpp - device code of the gfx950 rendering backend.
 *
 * What it computes: the reference's per-pixel pipeline `render0`
 * (core/tracer/tracer.cpp:1081-5405): primary rays, object-list traversal with
 * trnode transform caching and bounding-volume arrays, plane / quadric /
 * two-plane solvers, depth + axis + custom (CSG) clipping, Phong lighting with
 * hard shadows, refraction + Fresnel, reflection + metal/plain Fresnel,
 * FSAA reduce, gamma, 0x00RRGGBB packing.
 *
 * How it is organised for CDNA4 (this is not the reference's structure):
 *   - one lane = one ray (sample); a 64-lane wavefront = one 8x8 pixel block
 *     (4x4 / 8x4 pixels under 4x / 2x FSAA) and is its own workgroup; waves take
 *     entries of a host-computed schedule (footprints that can recurse first and
 *     with issue priority, the tile-list head in the entry, empty tiles leave at once).
 *   - COMPILED LISTS: the upload pass (qr_compile.cpp, qr_program.h) turns every list of the
 *     snapshot into a contiguous program of 32-byte cells whose opcode says what the walk does
 *     there (solver, which diff / ray it reads, cull, shadow class); the whole scene is one
 *     blob addressed by byte offsets from one base register.
 *   - WAVE-PACKET TRAVERSAL: lanes of a wave that walk the same list walk it
 *     together; the cell offset is wave-uniform, so cells and surface records are
 *     fetched with SCALAR loads (constant address space) into SGPRs and only per-ray
 *     quantities live in VGPRs.  Lanes with different lists (secondary rays leaving
 *     different surfaces) are served group by group (__ballot / readfirstlane) - the
 *     wave-level analogue of the reference's CHECK_MASK NONE/FULL packet early-outs
 *     (rtbase.h:1209).
 *   - per cell, a conservative bounding-sphere test (ours) lets the wave skip
 *     elements no ray can meet; bounding-volume arrays are skipped per ray and,
 *     when no ray of the group enters one, jumped over by the whole wave.
 *   - DEFERRED SHADING: the reference shades every hit that passes the depth
 *     test while it walks a list (overdraw); shading has no effect on the walk
 *     and fully overwrites the lane's colour, so walking first (keeping the
 *     depth-test sequence) and shading only the final hit is bit-identical and
 *     costs one shading (and one set of shadow rays) per ray
 *     (checked on the CPU by tests/test_oracle.py::test_deferred_shading_*).
 *   - recursion (context stack, tracer.h:426-665) becomes a per-lane explicit
 *     stack of 16-dword frames evaluated in the reference's order
 *     (refraction child, then reflection child), so colour arithmetic keeps the
 *     reference's association.
 *
 * Numeric contract: IEEE fp32, no contraction (-ffp-contract=off), correctly
 * rounded / and sqrt (hipcc default), compare predicates and integer
 * conversions as in oracle/qr_oracle.c's header.
 */
#ifndef QR_KERNEL_HPP
#define QR_KERNEL_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qr_scene.h"
#include "qr_program.h"

#ifndef QR_BLOCK
#define QR_BLOCK 64                /* one wave per workgroup: a wave slot is refilled the moment its wave ends (+8 % Mrays/s over 256) */
#endif
#ifndef QR_MAX_DEPTH
#define QR_MAX_DEPTH 10           /* RT_STACK_DEPTH, tracer.h:46 */
#endif
/* timing experiments (QR_DBG environment variable) exist only in -DQR_KNOBS builds */
#ifdef QR_KNOBS
#define QR_KNOB(bit) ((cx.dbg & (bit)) != 0)
#else
#define QR_KNOB(bit) false
#endif
#define QR_PER_LANE_TILE 0xFFFFFFFEu /* schedule entry: the footprint straddles tiles, look the list up per pixel (QR_SCHED_PER_LANE) */
#ifndef QR_RETURN_LOOP
#define QR_RETURN_LOOP 1     /* the instance with the per-lane walks: a lane unwinds all finished levels in one round (0: one level per round,
                              * as the packet instance does: there it costs demo1 1 % of its isolated launch, A/B in profiles/r04_return_loop_ab.txt) */
#endif
#ifndef QR_DYN_PRIO
#define QR_DYN_PRIO 1        /* issue priority follows the recursion round a wave is in (0: fixed by the footprint's class) */
#endif
#define QR_WT_SLOTS 14       /* QR_WAVETIME builds: u64 slots per wave */
#ifndef QR_MIN_WAVES_PER_SIMD
#define QR_MIN_WAVES_PER_SIMD 4   /* __launch_bounds__ 2nd argument: waves per SIMD */
#endif
#ifndef QR_DIVK_WAVES
#define QR_DIVK_WAVES 3           /* the instance with the per-lane walks: 168 VGPRs, none spilled.  (With walk_pool alone 4 waves and 70
                                   * spilled registers were 8 % faster; with the grid walk's state it is 93 spilled and 1 % slower.) */
#endif

typedef uint32_t u32;
#define QR_SMASK 0x80000000u

/* -DQR_PROF builds: wave-level event counters in a device global, printed by qr_render_count (where does a frame's instruction
 * budget go: tools/gpu_prof.py) */
#ifdef QR_PROF
__device__ unsigned long long qr_prof[64];
#define QR_PROF_HIT(i) do { if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) atomicAdd(&qr_prof[i], 1ull); } while (0)
#define QR_PROF_ADD(i, n) do { if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) atomicAdd(&qr_prof[i], (unsigned long long)(n)); } while (0)
/* algorithmic fp32 operations the kernel executes, with the weights of SURVEY.md 8(d) as oracle/qr_oracle.c applies them (FL()):
 * n operations for every lane that is active here / for every lane of `mask` */
#define QR_FLOPS(n) QR_PROF_ADD(48, (unsigned long long)(n) * (unsigned long long)__popcll(__ballot(true)))
#define QR_FLOPS_M(n, cnt) QR_PROF_ADD(48, (unsigned long long)(n) * (unsigned long long)(cnt))
/* beside them, the arithmetic of this implementation's own culls (no step of SURVEY 8(d): never part of a roofline numerator):
 * walk set-up 9 (+ 6 with box cells), sphere test 20, box slab test 22 */
#define QR_CULL_FLOPS(n) QR_PROF_ADD(49, (unsigned long long)(n) * (unsigned long long)__popcll(__ballot(true)))
#else
#define QR_PROF_HIT(i) do { } while (0)
#define QR_PROF_ADD(i, n) do { } while (0)
#define QR_FLOPS(n) do { } while (0)
#define QR_FLOPS_M(n, cnt) do { } while (0)
#define QR_CULL_FLOPS(n) do { } while (0)
#endif

/* what a launch needs besides the scene image: kernel arguments */
struct LaunchP
{
    const char *B;                      /* the compiled scene (qr_program.h), device memory                    */
    const uint32_t *order;              /* wave schedule, 8 B per wave: {bx | by << 14 | heaviness << 30, list offset} */
    int32_t n_blocks;
    int32_t depth;
    int32_t row_begin, row_end;         /* rows rendered by this launch                                        */
    int32_t index, thnum;               /* reference row interleave                                            */
    int32_t group_first, group_stride;  /* 8-row groups: first + k*stride                                      */
    unsigned long long *stats;          /* QR_STATS / QR_WAVETIME builds only                                  */
    int32_t dbg;                        /* timing experiments only (QR_DBG)                                    */
};

/* ------------------------------------------------------------------------ */
/* lane primitives (same definitions as the oracle)                          */
/* ------------------------------------------------------------------------ */

__device__ __forceinline__ u32   f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(u32 u)   { return __uint_as_float(u); }

__device__ __forceinline__ bool ceq(float a, float b) { return a == b; }
__device__ __forceinline__ bool cne(float a, float b) { return !(a == b); }
__device__ __forceinline__ bool clt(float a, float b) { return a < b; }
__device__ __forceinline__ bool cle(float a, float b) { return a <= b; }
__device__ __forceinline__ bool cgt(float a, float b) { return !(a <= b); }
__device__ __forceinline__ bool cge(float a, float b) { return !(a < b); }

__device__ __forceinline__ float fxor(float a, u32 m) { return u2f(f2u(a) ^ m); }
__device__ __forceinline__ float fabs_bits(float a)   { return u2f(f2u(a) & 0x7FFFFFFFu); }
__device__ __forceinline__ float rsq(float x) { return 1.0f / __builtin_sqrtf(x); }

__device__ __forceinline__ int32_t cvt_floor(float x)
{
    float f = __builtin_floorf(x);
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int32_t)f : (int32_t)0x80000000u;
}
__device__ __forceinline__ int32_t cvt_near(float x)
{
    float f = __builtin_rintf(x);
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int32_t)f : (int32_t)0x80000000u;
}

struct V3 { float x, y, z; };

__device__ __forceinline__ float vget(const V3 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
__device__ __forceinline__ void  vset(V3 &v, int i, float f)
{
    v.x = i == 0 ? f : v.x; v.y = i == 1 ? f : v.y; v.z = i == 2 ? f : v.z;
}

#define FLAG_SIDE 1
#define FLAG_PASS_THRU 2

/*
 * Wave-uniform records are read through the CONSTANT address space: with a
 * uniform (readfirstlane-derived) offset the backend then selects s_load_dword*
 * into SGPRs instead of 64 identical vector loads.  The image is never written
 * while a launch is in flight.
 */
#define QR_CONST __attribute__((address_space(4)))
typedef const QR_CONST char *BaseP;
typedef const QR_CONST DevHeader *FrmP;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x8 __attribute__((ext_vector_type(8)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ FrmP c_frm(BaseP B) { return (FrmP)B; }
#pragma clang diagnostic pop

/* per-wave context: the blob through both address spaces (uniform -> scalar loads, per-lane -> vector loads) */
struct Ctx
{
    BaseP B;
    const char *G;
    u32 off_shade;
    unsigned long long *stats;
    int dbg;
};

/* ------------------------------------------------------------------------ */
/* per-lane traversal state                                                  */
/* ------------------------------------------------------------------------ */

struct Ray
{
    V3 org, dir;            /* ctx_ORG, ctx_RAY_X..Z                          */
    float tmin, tmax;       /* ctx_T_MIN, initial ctx_T_BUF                   */
    u32 list;               /* byte offset of the list program, 0 = none      */
    u32 osrf;               /* ctx_PARAM(OBJ): byte offset of the originating surface's DSurf, 0 = none */
    int oflg;               /* ctx_PARAM(FLG) & 3: side | pass-thru           */
    V3 ploc;                /* parent's local hit (parent ctx_NRM_I..K)       */
};

struct Hit
{
    float t;                /* final ctx_T_BUF                                */
    u32 srf;                /* byte offset of the hit surface's DSurf, 0 = none */
    int side;
    V3 loc;                 /* local (possibly conic-adjusted) hit, ctx_NEW   */
};

#include "qr_walk.hpp"
#include "qr_shade.hpp"
#include "qr_pt_eager.hpp"

/* ------------------------------------------------------------------------ */
/* the kernel                                                                */
/* ------------------------------------------------------------------------ */

__device__ __forceinline__ float clamp1(float x) { return x < 1.0f ? x : 1.0f; }

/*
 * Path-tracer mode (RT_FEAT_PT; the third kernel instance only).  What the engine keeps per frame buffer
 * (engine.cpp:2875-2893): one LCG state and three colour planes per pixel sample; pts_o = 1 / frames so far,
 * pts_u = 1 - pts_o (tracer.cpp:1112-1136).
 */
struct PtParams { u32 *seeds; float *acc_r, *acc_g, *acc_b; float pts_o, pts_u; int eager, pad; };

/* one wave = one schedule entry: footprint `ord`, its tile-list program, rendered into `frame` */
/*
 * The pixel sample a lane stands for and whether this launch owns it, from the schedule word.  Computed where it is needed --
 * at the start of a wave, in the empty-tile exit, at the final store -- each time through opaque copies of its inputs: as
 * common subexpressions the coordinates were four registers live across the whole recursion, and the 128-register kernel
 * instance spilled them (2 KB of scratch traffic per wave: two thirds of the HBM bytes of the deep-recursion frames).
 */
__device__ __forceinline__ bool pixel_of(u32 ord, int fsaa, const LaunchP &lp, FrmP fr, int &x, int &y, int &k)
{
    asm volatile("" : "+s"(ord));
    int lane = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(lane));
    const int fw = fsaa == 2 ? 4 : 8, fh = fsaa == 0 ? 8 : 4;
    const int pix = lane >> fsaa;               /* pixel index inside the wave */
    k = lane & ((1 << fsaa) - 1);               /* sample index inside the pixel */
    const int px = fsaa == 2 ? (pix & 3) : (pix & 7), py = fsaa == 2 ? (pix >> 2) : (pix >> 3);
    x = (int)(ord & 0x3FFFu) * fw + px;
    y = (int)((ord >> 14) & 0x3FFFu) * fh + py;
    const int group = y >> 3;
    bool inside = x < fr->fr.frm_w && y < fr->fr.frm_h && y >= lp.row_begin && y < lp.row_end;
    if (group < lp.group_first) inside = false;
    if (lp.group_stride != 1 && (group - lp.group_first) % lp.group_stride != 0) inside = false;
    if (inside && lp.thnum > 1) inside = (y % lp.thnum) == lp.index;
    return inside;
}

template <bool COUNT, bool DIVK, bool PT = false>
__device__ __forceinline__ void render_wave(const LaunchP &lp, const u32 ord, const u32 sched_head, const int gw,
                                            uint32_t *__restrict__ frame, int32_t *__restrict__ ids,
                                            unsigned long long *__restrict__ counters, const PtParams *ptp = nullptr)
{
#ifdef QR_WAVETIME
    const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wt_clk0 = __builtin_amdgcn_s_memtime();      /* shader cycles: with the 100 MHz stamps, the clock the wave ran at */
    qr_wt_groups[0] = 0; qr_wt_groups[1] = 0; qr_wt_shadow = 0; qr_wt_cells[0] = qr_wt_cells[1] = qr_wt_cells[2] = qr_wt_cells[3] = 0;
    unsigned long long wt_mid = 0, wt_trav = 0, wt_shade = 0, wt_t0 = 0; u32 wt_push = 0;
#endif
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    Ctx cx;
    cx.B = (BaseP)lp.B; cx.G = lp.B; cx.stats = lp.stats; cx.dbg = lp.dbg;
#pragma clang diagnostic pop
    const BaseP B = cx.B;
    const FrmP fr = c_frm(B);
    cx.off_shade = fr->off_shade;
    const int fsaa = fr->fr.fsaa;
    const int ns = 1 << fsaa;
    const int lane = threadIdx.x & 63;
    const int pix = lane >> fsaa;               /* pixel index inside the wave */
    const int k = lane & (ns - 1);              /* sample index inside the pixel */

    /* wave footprint: 8x8 pixels (no AA), 8x4 (2x), 4x4 (4x).  Every WAVE takes one entry of
     * the host-computed schedule (footprints that can spawn deep recursion first, so that their
     * long waves overlap the bulk instead of forming a tail); consecutive entries are
     * neighbouring footprints, so waves running side by side still share tile lists in the scalar cache. */
    const int fw = fsaa == 2 ? 4 : 8, fh = fsaa == 0 ? 8 : 4;
    (void)gw;
    /* footprints that can recurse get issue priority: the frame ends with the slowest of them, and while
     * the bulk is in flight they would otherwise share their SIMD's issue slots evenly */
#if QR_DYN_PRIO
    /* ... and among them the ones that actually do: every traversal round a wave starts raises its priority (2 waves of demo
     * scene 1 at 1080p run 7 rounds, 20 run 6, 12 000 one -- but 3 900 footprints CAN recurse and fill the first generation) */
    if (ord >> 30) __builtin_amdgcn_s_setprio(1);
    int prio_round = 0;
#else
    if (ord >> 30) { if ((ord >> 30) >= 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
#endif
    const int px = fsaa == 2 ? (pix & 3) : (pix & 7), py = fsaa == 2 ? (pix >> 2) : (pix >> 3);
    const int x = (int)(ord & 0x3FFFu) * fw + px;
    const int y = (int)((ord >> 14) & 0x3FFFu) * fh + py;
    const int group = y >> 3;
    const int frm_w = fr->fr.frm_w;

    bool inside = x < frm_w && y < fr->fr.frm_h && y >= lp.row_begin && y < lp.row_end;
    if (group < lp.group_first) inside = false;
    if (lp.group_stride != 1 && (group - lp.group_first) % lp.group_stride != 0) inside = false;
    if (inside && lp.thnum > 1) inside = (y % lp.thnum) == lp.index;
    if (!any_lane(inside)) return;                 /* whole wave outside this launch's rows */
    if (sched_head < 256u)
    {
        /* empty tiles: no ray of these footprints meets anything; the reference's pipeline ends with colour 0 for such a
         * packet (clamp, sqrt and cvt of 0 are 0), so store it and leave.  The head is the number of footprints of the
         * run along the row (qr_compile.cpp; 0: one) */
        const u32 run = sched_head == 0 ? 1u : sched_head;
        unsigned long long n = 0;
        for (u32 i = 0; i < run; i++)
        {
            int x0, y0, k0;
            const bool in0 = pixel_of(ord + i, fsaa, lp, fr, x0, y0, k0);
            if (in0 && k0 == 0)
            {
                frame[(size_t)y0 * frm_w + x0] = 0u;
                if (ids != nullptr) ids[(size_t)y0 * frm_w + x0] = -1;
            }
            if (COUNT && in0) n++;
        }
        if (COUNT)
        {
            for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
            if (lane == 0 && n != 0) atomicAdd(&counters[0], n);
        }
        return;
    }

    Counters cnt = {0, 0, 0, 0};
    const float t_inf = fr->fr.t_max;

    u32 rng = 0;                                /* PT: this sample's LCG state */
    /* primary ray, tracer.cpp:1287-1322; sample offsets engine.cpp:3480-3550 */
    Ray ray;
    {
        int ai = 0;
        if (fsaa == 1) ai = (x & 1) * 2 + k;
        if (fsaa == 2) ai = k;
        float ha, va;
        if (fsaa == 0) { ha = fr->fr.hor_a[0]; va = fr->fr.ver_a[0]; }       /* wave-uniform: scalar loads */
        else
        {
            const qr_frame *gf = (const qr_frame *)cx.G;
            ha = gf->hor_a[ai]; va = gf->ver_a[ai];
        }
        float hr = 0.0f, vr = 0.0f;
        if constexpr (PT)
        {
            /* tent-filter jitter of the sample position, tracer.cpp:1218-1285 */
            if (inside && ptp->eager != 3)
            {
                rng = ptp->seeds[((size_t)y * fr->fr.frm_row + x) * ns + k];     /* the engine's slot: row stride frm_row (tracer.cpp:1168-1176) */
                float a = pt_random(rng); a = a + a;
                hr = a < 1.0f ? __builtin_sqrtf(a) - 1.0f : 1.0f - __builtin_sqrtf(2.0f - a);
                float b = pt_random(rng); b = b + b;
                vr = b < 1.0f ? __builtin_sqrtf(b) - 1.0f : 1.0f - __builtin_sqrtf(2.0f - b);
                hr = hr * 0.5f; vr = vr * 0.5f;
                if (fsaa != 0) { hr = hr * 0.5f; vr = vr * 0.5f; }
            }
        }
        float hs = (float)x + ha; hs = hs + hr;
        float vs = (float)y + va; vs = vs + vr;
        float x1 = fr->fr.hor[0] * hs, x2 = fr->fr.hor[1] * hs, x3 = fr->fr.hor[2] * hs;
        float x4 = fr->fr.ver[0] * vs, x5 = fr->fr.ver[1] * vs, x6 = fr->fr.ver[2] * vs;
        x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
        ray.dir.x = x1 + fr->fr.dir[0];
        ray.dir.y = x2 + fr->fr.dir[1];
        ray.dir.z = x3 + fr->fr.dir[2];
        ray.org.x = fr->fr.org[0]; ray.org.y = fr->fr.org[1]; ray.org.z = fr->fr.org[2];
        ray.tmin = fr->fr.t_min; ray.tmax = t_inf;
        ray.osrf = 0; ray.oflg = 0;
        ray.ploc = {0, 0, 0};
        ray.list = 0;
        if (sched_head != QR_PER_LANE_TILE)
        {
            if (inside) ray.list = sched_head;
        }
        else if (inside)
        {
            const int tile = (y / fr->fr.tile_h) * fr->fr.tls_row + (x / fr->fr.tile_w);
            ray.list = *(const u32 *)(cx.G + (fr->off_tiles + (u32)tile * 4u));
        }
    }

    /* recursion frames: levels 0..LDSL-1 in LDS ([level][quarter][lane]: a quarter of all lanes is contiguous,
     * 16-byte accesses at a 16-byte lane stride), deeper levels in scratch */
    /* Ray tracer: a frame's first two quarters (colour so far, factors: all a node with ONE child ever reads back) live in LDS
     * for the four shallowest levels and in scratch below; the other two (the second child's origin, direction and local hit:
     * only written by a node with BOTH children) in scratch.  8 KB of LDS per wave as before, but of the frames demo scene 2
     * pushes at 3840x2160 with 4x FSAA -- 18 M with one child, 9 M with two, a third of them deeper than level 1 -- most never
     * touch scratch now.  Path tracer (six quarters: its bounce): one whole level in LDS. */
    constexpr int LDSL = PT ? 1 : 0;
    constexpr int NLDS = PT ? 0 : QR_LDS_NARROW_LEVELS;
    constexpr int FQ = PT ? 6 : 4;              /* quarters per frame: the path tracer also keeps its bounce (q4, q5) */
    __shared__ f32x4 lds_frames[LDSL > 0 ? LDSL : 1][FQ][PT ? 64 : 1];
    __shared__ f32x4 lds_narrow[NLDS > 0 ? NLDS : 1][2][PT ? 1 : 64];
    f32x4 deep[PT ? QR_MAX_DEPTH - LDSL : 1][FQ];
    f32x4 deep_narrow[PT ? 1 : QR_MAX_DEPTH - NLDS][2];
    f32x4 deep_wide[PT ? 1 : QR_MAX_DEPTH][2];
    auto frame_put = [&](int level, int quarter, f32x4 v) {
        if constexpr (PT) { if (level < LDSL) lds_frames[level][quarter][lane] = v; else deep[level - LDSL][quarter] = v; }
        else if (quarter >= 2) deep_wide[level][quarter - 2] = v;
        else if (level < NLDS) lds_narrow[level][quarter][lane] = v;
        else deep_narrow[level - NLDS][quarter] = v;
    };
    auto frame_get = [&](int level, int quarter) -> f32x4 {
        if constexpr (PT) return level < LDSL ? lds_frames[level][quarter][lane] : deep[level - LDSL][quarter];
        else if (quarter >= 2) return deep_wide[level][quarter - 2];
        else return level < NLDS ? lds_narrow[level][quarter][lane] : deep_narrow[level - NLDS][quarter];
    };
    Outer ou;
    ou.sp = 0;
    ou.mode = inside ? 0 : 2;                   /* 0 trace, 1 return, 2 done */
    ou.ret = {0, 0, 0};
    ou.hit_id = -1;
    /* the primary hit id is only wanted by id renders and only when the wave ends: it waits in LDS, not in a register that
     * every frame's waves would carry through the whole recursion */
    __shared__ int lds_hit_id[64];
    if (ids != nullptr) lds_hit_id[lane] = -1;
    int &sp = ou.sp, &mode = ou.mode;
    V3 &ret = ou.ret;
    const int depth = lp.depth;

    if (COUNT && inside) cnt.primary++;
    QR_FLOPS_M(16, __popcll(__ballot(inside)));             /* primary ray */

    bool eager_done = false;
    if constexpr (PT)
    {
        if (ptp->eager)
        {
            ret = pt_eager(cx, ray, inside, depth, t_inf, rng, ptp->eager == 3);
            eager_done = true;
        }
    }

    while (!eager_done && any_lane(mode != 2))
    {
        const bool tr = mode == 0;
        bool ret_once = true;
#ifdef QR_STATS
        {
            const unsigned long long n_tr = (unsigned long long)__popcll(__ballot(tr)), n_on = (unsigned long long)__popcll(__ballot(mode != 2));
            if (__ffsll((long long)__ballot(true)) - 1 == lane) { atomicAdd(&cx.stats[28], 1ull); atomicAdd(&cx.stats[29], n_tr); atomicAdd(&cx.stats[30], n_on); }
        }
#endif
        if (any_lane(tr))
        {
            Hit h; bool occ;
#ifdef QR_WAVETIME
            wt_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#if QR_DYN_PRIO
            if (prio_round < 3) { prio_round++; if (prio_round == 2) __builtin_amdgcn_s_setprio(2); else if (prio_round == 3) __builtin_amdgcn_s_setprio(3); }
#endif
            /* coherent: every ray of this round is a primary ray (neighbouring pixels) */
            const bool coherent = !any_lane(tr && sp != 0);
            traverse<false, DIVK>(B, tr, coherent, ray, h, occ
#ifdef QR_STATS
                            , cx.stats
#endif
                            );
#ifdef QR_WAVETIME
            if (wt_mid == 0) wt_mid = __builtin_amdgcn_s_memrealtime();
            wt_push++;
            wt_trav += __builtin_amdgcn_s_memrealtime() - wt_t0;
            wt_t0 = __builtin_amdgcn_s_memrealtime();
#endif
            const bool got = tr && h.srf != 0 && !QR_KNOB(4);
            if (tr && !got) { ret = {0, 0, 0}; mode = 1; }
            const int hsi = (int)((h.srf - QR_OFF_SRF) >> 7);          /* surface index: DSurf records are 128 B */
            if (ids != nullptr)
            {
                int lane_h = (int)(threadIdx.x & 63u);
                asm volatile("" : "+v"(lane_h));
                if (got && sp == 0) lds_hit_id[lane_h] = (hsi << 1) | h.side;
            }

            Shaded o;
            shade<COUNT, DIVK, PT>(cx, got, coherent, ray, h, o, cnt, &rng, depth - sp);
#ifdef QR_WAVETIME
            wt_shade += __builtin_amdgcn_s_memrealtime() - wt_t0;
#endif

            if (got)
            {
                const bool can_spawn = (depth - sp) != 0;
                /* PT: one more flag in front of the surface index: the node has a diffuse bounce to follow LAST (the
                 * node's colour is linear in its children: rfl c_rfl + trn c_trn + x0 (bounce l_dff tex + emission)) */
                const bool has_pt = PT && o.want_pt && can_spawn;
                const int meta = PT ? ((hsi << 5) | (has_pt ? 16 : 0) | (h.side << 3) | (o.want_rf ? 4 : 0))
                                    : ((hsi << 4) | (h.side << 3) | (o.want_rf ? 4 : 0));
                if constexpr (PT)
                {
                    if (has_pt)
                    {
                        frame_put(sp, 4, f32x4{o.ptw.x * o.x0, o.ptw.y * o.x0, o.ptw.z * o.x0, 0.0f});
                        frame_put(sp, 5, f32x4{o.pdir.x, o.pdir.y, o.pdir.z, 0.0f});
                    }
                }
#ifdef QR_PROF
                {
                    /* frames pushed, by level (lanes): wide = both children, narrow = one */
                    const bool wide = o.want_tr && can_spawn && (o.want_rf || has_pt), narrow = can_spawn && !wide && (o.want_tr || o.want_rf);
                    for (int lv = 0; lv < 8; lv++)
                    {
                        const unsigned long long nw = __popcll(__ballot(wide && (sp < 7 ? sp : 7) == lv)), nn = __popcll(__ballot(narrow && (sp < 7 ? sp : 7) == lv));
                        if (nw) QR_PROF_ADD(32 + lv, nw);
                        if (nn) QR_PROF_ADD(40 + lv, nn);
                    }
                }
#endif
                if (o.want_tr && can_spawn)
                {
                    frame_put(sp, 0, f32x4{o.col.x, o.col.y, o.col.z, __int_as_float(meta | 1)});
                    frame_put(sp, 1, f32x4{o.c_trn, o.c_rfl, o.x0, o.hit.x});
                    if (o.want_rf || has_pt)
                    {
                        /* only a node that also has a reflection child needs its direction and local hit later */
                        frame_put(sp, 2, f32x4{o.rdir.x, o.rdir.y, o.rdir.z, o.hit.y});
                        frame_put(sp, 3, f32x4{o.loc.x, o.loc.y, o.loc.z, o.hit.z});
                    }
                    sp++;
                    ray.org = o.hit; ray.dir = o.tdir; ray.tmin = 0.0f; ray.tmax = t_inf;
                    ray.list = o.lst_tr; ray.osrf = h.srf; ray.oflg = h.side | FLAG_PASS_THRU;
                    ray.ploc = o.loc;
                    mode = 0;
                    if (COUNT) cnt.refract++;
                }
                else
                {
                    /* TR_mix with a zero child colour, 3560-3598 */
                    V3 c;
                    c.x = 0.0f + o.col.x * o.x0;
                    c.y = 0.0f + o.col.y * o.x0;
                    c.z = 0.0f + o.col.z * o.x0;
                    if (o.want_rf && can_spawn)
                    {
                        /* a reflection-only frame is read back for its colour and factor alone */
                        frame_put(sp, 0, f32x4{c.x, c.y, c.z, __int_as_float(meta | 2)});
                        frame_put(sp, 1, f32x4{o.c_trn, o.c_rfl, o.x0, o.hit.x});
                        if (has_pt)
                        {
                            frame_put(sp, 2, f32x4{0.0f, 0.0f, 0.0f, o.hit.y});
                            frame_put(sp, 3, f32x4{o.loc.x, o.loc.y, o.loc.z, o.hit.z});
                        }
                        sp++;
                        ray.org = o.hit; ray.dir = o.rdir; ray.tmin = 0.0f; ray.tmax = t_inf;
                        ray.list = o.lst_rf; ray.osrf = h.srf; ray.oflg = h.side;
                        ray.ploc = o.loc;
                        mode = 0;
                        if (COUNT) cnt.reflect++;
                    }
                    else
                    {
                        if (o.want_rf) { c.x = 0.0f + c.x; c.y = 0.0f + c.y; c.z = 0.0f + c.z; }
                        if (has_pt)
                        {
                            /* no other child: the bounce at once */
                            frame_put(sp, 0, f32x4{c.x, c.y, c.z, __int_as_float(meta | 3)});
                            sp++;
                            ray.org = o.hit; ray.dir = o.pdir; ray.tmin = 0.0f; ray.tmax = t_inf;
                            ray.list = o.lst_pt; ray.osrf = h.srf; ray.oflg = h.side;
                            ray.ploc = o.loc;
                            mode = 0;
                        }
                        else
                        {
                        ret = c;
                        mode = 1;
                        }
                    }
                }
            }
        }

        /* returns.  DIVK instance: a lane unwinds until its sample is finished or has its next ray (one level per round before
         * round 4) -- a lane's state machine does not depend on its neighbours, and every unfinished lane then traces in every
         * round: fewer, fuller rounds (config 5: +1.3 %).  The packet instance keeps one level per round */
        while ((QR_RETURN_LOOP && DIVK) ? mode == 1 : (mode == 1 && ret_once))
        {
            ret_once = false;
            if (sp == 0)
            {
                mode = 2;
            }
            else
            {
                const f32x4 q0 = frame_get(sp - 1, 0), q1 = frame_get(sp - 1, 1);
                const int fmeta = __float_as_int(q0.w);
                const int phase = fmeta & 3;
                constexpr int MS = PT ? 5 : 4;              /* surface index sits above the flags */
                /* PT: the node's other children are done, `c` is its colour without the bounce: follow the bounce now */
                auto bounce = [&](V3 c) {
                    const f32x4 q2 = frame_get(sp - 1, 2), q3 = frame_get(sp - 1, 3), q5 = frame_get(sp - 1, 5);
                    frame_put(sp - 1, 0, f32x4{c.x, c.y, c.z, __int_as_float(fmeta | 3)});
                    const int psi = fmeta >> MS, pside = (fmeta >> 3) & 1;
                    ray.org = {q1.w, q2.w, q3.w};
                    ray.dir = {q5.x, q5.y, q5.z};
                    ray.tmin = 0.0f; ray.tmax = t_inf;
                    ray.list = ((const DShade *)(cx.G + (cx.off_shade + (u32)psi * (u32)sizeof(DShade))))->lst[pside];
                    ray.osrf = QR_OFF_SRF + ((u32)psi << 7); ray.oflg = pside;
                    ray.ploc = {q3.x, q3.y, q3.z};
                    mode = 0;
                };
                if (PT && phase == 3)
                {
                    /* PT_ret 2598-2620: the bounce's colour times l_dff tex (times x0, see above) */
                    const f32x4 q4 = frame_get(sp - 1, 4);
                    ret.x = q0.x + ret.x * q4.x;
                    ret.y = q0.y + ret.y * q4.y;
                    ret.z = q0.z + ret.z * q4.z;
                    sp--;
                }
                else if (phase == 1)
                {
                    /* TR_ret + TR_mix 3534-3598 */
                    V3 c;
                    c.x = ret.x * q1.x + q0.x * q1.z;
                    c.y = ret.y * q1.x + q0.y * q1.z;
                    c.z = ret.z * q1.x + q0.z * q1.z;
                    if (fmeta & 4)
                    {
                        /* reflection child of the same node (depth budget is
                         * the same as for the refraction child) */
                        const f32x4 q2 = frame_get(sp - 1, 2), q3 = frame_get(sp - 1, 3);
                        frame_put(sp - 1, 0, f32x4{c.x, c.y, c.z, __int_as_float((fmeta & ~3) | 2)});
                        const int psi = fmeta >> MS, pside = (fmeta >> 3) & 1;
                        ray.org = {q1.w, q2.w, q3.w};
                        ray.dir = {q2.x, q2.y, q2.z};
                        ray.tmin = 0.0f; ray.tmax = t_inf;
                        ray.list = ((const DShade *)(cx.G + (cx.off_shade + (u32)psi * (u32)sizeof(DShade))))->lst[pside];
                        ray.osrf = QR_OFF_SRF + ((u32)psi << 7); ray.oflg = pside;
                        ray.ploc = {q3.x, q3.y, q3.z};
                        mode = 0;
                        if (COUNT) cnt.reflect++;
                    }
                    else if (PT && (fmeta & 16)) bounce(c);
                    else
                    {
                        ret = c;
                        sp--;
                    }
                }
                else
                {
                    /* RF_ret + RF_mix 3868-3908 */
                    V3 c;
                    c.x = ret.x * q1.y + q0.x;
                    c.y = ret.y * q1.y + q0.y;
                    c.z = ret.z * q1.y + q0.z;
                    if (PT && (fmeta & 16)) bounce(c);
                    else
                    {
                        ret = c;
                        sp--;
                    }
                }
            }
        }
    }

#ifdef QR_WAVETIME
    if (!COUNT && __ffsll((long long)__ballot(true)) - 1 == lane)
    {
        unsigned long long *o = counters + 64 + (size_t)gw * QR_WT_SLOTS;
        o[0] = wt_start; o[1] = wt_mid; o[2] = __builtin_amdgcn_s_memrealtime();
        o[12] = wt_clk0; o[13] = __builtin_amdgcn_s_memtime();
        o[4] = ord; o[5] = qr_wt_groups[0]; o[6] = qr_wt_groups[1]; o[7] = wt_push; o[8] = wt_trav; o[9] = wt_shade; o[10] = qr_wt_shadow;
        o[11] = (unsigned long long)qr_wt_cells[0] | ((unsigned long long)qr_wt_cells[1] << 16) | ((unsigned long long)qr_wt_cells[2] << 32) | ((unsigned long long)qr_wt_cells[3] << 48);
        o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))
             | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 32)
             | ((unsigned long long)wt_push << 40);
    }
#endif
    if constexpr (PT)
    {
        /* 5176-5219: running mean of the samples in the colour planes; the frame shows the mean so far */
        if (inside)
        {
            const size_t si = ((size_t)y * fr->fr.frm_row + x) * ns + k;
            ptp->seeds[si] = rng;
            const float ar = ret.x * ptp->pts_o + ptp->acc_r[si] * ptp->pts_u;
            const float ag = ret.y * ptp->pts_o + ptp->acc_g[si] * ptp->pts_u;
            const float ab = ret.z * ptp->pts_o + ptp->acc_b[si] * ptp->pts_u;
            ptp->acc_r[si] = ar; ptp->acc_g[si] = ag; ptp->acc_b[si] = ab;
            ret = {ar, ag, ab};
        }
    }
    QR_FLOPS_M(6 + 2 * fsaa, __popcll(__ballot(inside)));
    /* XX_end 5161-5343: clamp, FSAA reduce, gamma, pack */
    float cr = clamp1(ret.x), cg = clamp1(ret.y), cb = clamp1(ret.z);
    if (fsaa >= 1)
    {
        cr = cr * 0.5f; cg = cg * 0.5f; cb = cb * 0.5f;
        cr = cr + __shfl_down(cr, 1); cg = cg + __shfl_down(cg, 1); cb = cb + __shfl_down(cb, 1);
    }
    if (fsaa >= 2)
    {
        cr = cr * 0.5f; cg = cg * 0.5f; cb = cb * 0.5f;
        cr = cr + __shfl_down(cr, 2); cg = cg + __shfl_down(cg, 2); cb = cb + __shfl_down(cb, 2);
    }
    /* pixel coordinates and row ownership once more (pixel_of) */
    int x_e, y_e, k_e;
    const bool inside_e = pixel_of(ord, fsaa, lp, fr, x_e, y_e, k_e) && k_e == 0;
    if (inside_e)
    {
        if (fr->fr.ctx_flags & QR_PROP_GAMMA)
        {
            asm volatile("" ::: "memory");      /* keep the branch: three IEEE square roots are not worth speculating */
            cr = __builtin_sqrtf(cr); cg = __builtin_sqrtf(cg); cb = __builtin_sqrtf(cb);
        }
        const float cl = fr->fr.clamp; const u32 cmask = fr->fr.cmask;
        cr = cr * cl; cg = cg * cl; cb = cb * cl;
        const u32 p = (((u32)cvt_near(cr) & cmask) << 16) |
                      (((u32)cvt_near(cg) & cmask) << 8) |
                       ((u32)cvt_near(cb) & cmask);
        frame[(size_t)y_e * frm_w + x_e] = p;
        if (ids != nullptr)
        {
            int lane_e = (int)(threadIdx.x & 63u);
            asm volatile("" : "+v"(lane_e));        /* not the address register of the wave's first instructions, kept alive */
            ids[(size_t)y_e * frm_w + x_e] = lds_hit_id[lane_e];
        }
    }

    if (COUNT)
    {
        /* one atomic per wave and counter */
        unsigned long long v[4] = { cnt.primary, cnt.shadow, cnt.reflect, cnt.refract };
        for (int i = 0; i < 4; i++)
        {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
            if (lane == 0 && s != 0) atomicAdd(&counters[i], s);
        }
    }
}

/* single-scene launch */
template <bool COUNT, int WAVES, bool DIVK>
__global__ __launch_bounds__(QR_BLOCK, WAVES)
void qr_render_kernel(LaunchP lp, uint32_t *__restrict__ frame, int32_t *__restrict__ ids,
                      unsigned long long *__restrict__ counters)
{
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (QR_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (gw >= lp.n_blocks) return;
    /* schedule entry {footprint coordinates, tile-list offset or QR_PER_LANE_TILE}: one scalar load */
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const u32x2 sched = ((const QR_CONST u32x2 *)lp.order)[gw];
#pragma clang diagnostic pop
    render_wave<COUNT, DIVK>(lp, sched.x, sched.y, gw, frame, ids, counters);
}

/* path-tracer launch: one sample per pixel sample and call, accumulated in the planes of `pt` */
__global__ __launch_bounds__(QR_BLOCK, 3)
void qr_render_pt_kernel(LaunchP lp, PtParams pt, uint32_t *__restrict__ frame, unsigned long long *__restrict__ counters)
{
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (QR_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (gw >= lp.n_blocks) return;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const u32x2 sched = ((const QR_CONST u32x2 *)lp.order)[gw];
#pragma clang diagnostic pop
    render_wave<false, false, true>(lp, sched.x, sched.y, gw, frame, nullptr, counters, &pt);
}

/*
 * Multi-target launch (qr_render_multi_async): ONE grid renders row ranges of several frames -- of the
 * same or of different scenes -- so that the N blocks a GPU owns in a sharded step share one launch: a
 * frame cut into N small launches pays N ramps, drains and tails (0.23 ms for 8 blocks of demo1 at
 * 1080p against 0.09 ms for the whole frame).  Schedule entries are 16 bytes {footprint, tile-list
 * offset, target index, 0}, heavy footprints of all targets first.  Everything a target needs travels
 * in the kernel arguments (nothing of a scene's launch state is cached on the device).
 */
#define QR_MAX_TARGETS 16
struct DevTarget { uint32_t *frame; const char *B; int32_t row_begin, row_end; int32_t depth, pad; };
struct DevTargets { DevTarget t[QR_MAX_TARGETS]; };

template <int WAVES, bool DIVK>
__global__ __launch_bounds__(QR_BLOCK, WAVES)
void qr_render_multi_kernel(DevTargets tg, const uint32_t *__restrict__ order16, int n_blocks,
                            unsigned long long *__restrict__ counters)
{
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (QR_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (gw >= n_blocks) return;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const u32x4 sched = ((const QR_CONST u32x4 *)order16)[gw];
#pragma clang diagnostic pop
    const DevTarget t = tg.t[sched.z & (QR_MAX_TARGETS - 1)];
    LaunchP lp;
    lp.B = t.B; lp.order = nullptr; lp.n_blocks = n_blocks; lp.depth = t.depth;
    lp.row_begin = t.row_begin; lp.row_end = t.row_end;
    lp.index = 0; lp.thnum = 1;
    lp.group_first = t.row_begin / 8; lp.group_stride = 1;
#ifdef QR_STATS
    lp.stats = counters + 4;            /* instrumented builds count into the first scene's counter block */
#else
    lp.stats = nullptr;
#endif
    lp.dbg = 0;
    render_wave<false, DIVK>(lp, sched.x, sched.y, gw, t.frame, nullptr, counters);
}

#endif /* QR_KERNEL_HPP */
