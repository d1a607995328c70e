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
 *   - WAVE-PACKET TRAVERSAL: lanes of a wave that walk the same list walk it
 *     together; the element index is wave-uniform, so list cells and surface
 *     records are fetched with SCALAR loads (constant address space) into
 *     SGPRs and only per-ray quantities live in VGPRs.  Lanes with different
 *     lists (secondary rays leaving different surfaces) are served group by
 *     group (__ballot / readfirstlane) - the wave-level analogue of the
 *     reference's CHECK_MASK NONE/FULL packet early-outs (rtbase.h:1209).
 *   - per cell, a conservative bounding-sphere test (ours) lets the wave skip
 *     elements no ray can meet; bounding-volume arrays are skipped per ray and,
 *     when no ray of the group enters one, jumped over by the whole wave.
 *   - a DIVERGENT variant (walk_div) walks every lane's list independently with
 *     vector loads, for scenes of thousands of small objects.
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

#ifndef QR_BLOCK
#define QR_BLOCK 64                /* one wave per workgroup: a wave slot is refilled the moment its wave ends (+8 % Mrays/s over 256) */
#endif
#ifndef QR_MAX_DEPTH
#define QR_MAX_DEPTH 10           /* RT_STACK_DEPTH, tracer.h:46 */
#endif
/* timing experiments (QR_DBG environment variable) exist only in -DQR_KNOBS builds: every knob the
 * production kernel tests costs a hoisted SGPR pair, and the allocator is already spilling SGPRs */
#ifdef QR_KNOBS
#define QR_KNOB(bit) ((sc.dbg & (bit)) != 0)
#else
#define QR_KNOB(bit) false
#endif
#define QR_PER_LANE_TILE (-2) /* schedule entry: the footprint straddles tiles, look the list up per pixel */
#define QR_WT_SLOTS 12       /* QR_WAVETIME builds: u64 slots per wave */
#ifndef QR_MIN_WAVES_PER_SIMD
#define QR_MIN_WAVES_PER_SIMD 4   /* __launch_bounds__ 2nd argument: waves per SIMD */
#endif

typedef uint32_t u32;

/*
 * Device-side surface record (built by qr_scene_upload from qr_surface):
 * 32 dwords, the first 20 ("hot") are everything the list walk needs for a
 * surface without transform; one s_load_dwordx16 + one s_load_dwordx4.
 */
struct DSurf
{
    float pos[3]; u32 flags;        /*  0 */
    float sci[4];                   /*  4 */
    float scj[3]; int32_t clip;     /*  8 */
    float min[3]; float d_eps;      /* 12 */
    float max[3]; float t_eps;      /* 16 */
    float tci[3]; int32_t trnode;   /* 20 */
    float tcj[3]; int32_t props0;   /* 24 */
    float tck[3]; int32_t props1;   /* 28 */
};

/* flags word of DSurf */
#define DF_MINMAX(f)  ((f) & 63u)
#define DF_CONIC(f)   (((f) >> 6) & 3u)
#define DF_TRM(f)     (((f) >> 8) & 3u)
#define DF_SHIFT(f)   (((f) >> 10) & 1u)
#define DF_MAP(f, n)  (((f) >> (11 + 2 * (n))) & 3u)
#define DF_SGN(f, n)  ((((f) >> (17 + (n))) & 1u) ? 0x80000000u : 0u)
#define DF_SOLVER(f)  (((f) >> 20) & 3u)
#define DF_NKIND(f)   (((f) >> 22) & 3u)
#define DF_CKIND(f)   (((f) >> 24) & 3u)
#define DF_ARRAY(f)   (((f) >> 26) & 1u)    /* tag < 0: array / trnode element */
#define DF_CDEF(f)    (((f) >> 28) & 1u)
#define QR_SMASK 0x80000000u

/* per-surface data only shading needs */
struct DShade
{
    int32_t mat[2];
    int32_t lst[4];
    int32_t pad[2];
};

/* device list cell: the snapshot's qr_elem plus the conservative world-space bounding sphere
 * of the cell's surface (one 32-byte scalar load serves both the walk and its cull test) */
struct DCell
{
    int32_t simd, data, next, kind;     /* kind bit 2: cullable (finite bound, plain surface cell) */
    float cx, cy, cz, r;
};

struct DevScene
{
    const DSurf       *__restrict__ srf;
    const DShade      *__restrict__ shd;
    const qr_material *__restrict__ mat;
    const qr_light    *__restrict__ lgt;
    const DCell       *__restrict__ elm;    /* list cells, 32 B: {simd,data,next,kind | bounding sphere} */
    const int32_t     *__restrict__ tiles;
    const uint32_t    *__restrict__ texels;
    const qr_frame    *__restrict__ frp;    /* frame/camera parameters (device memory, scalar-loaded on demand) */
    int32_t depth;
    int32_t row_begin, row_end;   /* rows rendered by this launch            */
    int32_t index, thnum;         /* reference row interleave                */
    int32_t group_first, group_stride, n_groups; /* 8-row groups: first + k*stride */
    const void        *__restrict__ bsph;   /* float4 per surface: world-space bounding sphere (cx,cy,cz,r) */
    const uint32_t    *__restrict__ order;  /* wave schedule, 8 B per wave: {bx | by << 14, tile-list head}, heavy footprints first */
    int32_t n_blocks;
    int32_t nested;             /* every surface list's arrays are properly nested (checked at upload) */
    unsigned long long *stats;    /* QR_STATS builds only: walk statistics */
    int32_t dbg;                  /* timing experiments only (QR_DBG): 1 no shadow walks, 2 no lights */
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

/*
 * Lane mask in a VGPR (0 / ~0), the reference's own representation.  Chains of `bool && bool`
 * compile to v_cmp -> s_and_b64 chains through SGPR pairs, each link a VALU->SALU round trip
 * (measured: ~20 cycles per link, the depth/min-max tests of CC_clp took ~570 cycles); as VGPR
 * words the same logic is v_cmp + v_cndmask + v_and.  The empty asm keeps the optimiser from
 * folding the words back into i1 logic.
 */
#ifdef QR_LM_FOLD
__device__ __forceinline__ u32 LM(bool c) { return c ? 0xFFFFFFFFu : 0u; }
#else
__device__ __forceinline__ u32 LM(bool c) { u32 x = c ? 0xFFFFFFFFu : 0u; asm("" : "+v"(x)); return x; }
#endif

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
 * uniform (readfirstlane-derived) index the backend then selects s_load_dword*
 * into SGPRs instead of 64 identical vector loads.  The arrays are never written
 * while a launch is in flight.
 */
#define QR_CONST __attribute__((address_space(4)))
typedef const QR_CONST DSurf   *SrfP;
typedef const QR_CONST DCell *ElmP;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ SrfP c_srf(const DevScene &sc) { return (SrfP)sc.srf; }
__device__ __forceinline__ ElmP c_elm(const DevScene &sc) { return (ElmP)sc.elm; }
typedef const QR_CONST qr_frame *FrmP;
__device__ __forceinline__ FrmP c_frm(const DevScene &sc) { return (FrmP)sc.frp; }
#pragma clang diagnostic pop

__device__ __forceinline__ qr_elem ld_elem(ElmP p)
{
    qr_elem e;
    e.simd = p->simd; e.data = p->data; e.next = p->next; e.kind = p->kind;
    return e;
}
struct CellS { qr_elem el; float cx, cy, cz, r; };
typedef u32 u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ CellS ld_cell(ElmP p)
{
    /* one 32-byte load: as two 16-byte halves the compiler sinks the sphere's half behind the test of
     * `kind`, which makes two dependent scalar-cache round trips per cell */
    typedef u32 u32x8_ __attribute__((ext_vector_type(8)));
    const u32x8_ a = *(const QR_CONST u32x8_ *)p;
    CellS c;
    c.el.simd = (int)a.s0; c.el.data = (int)a.s1; c.el.next = (int)a.s2; c.el.kind = (int)a.s3;
    c.cx = u2f(a.s4); c.cy = u2f(a.s5); c.cz = u2f(a.s6); c.r = u2f(a.s7);
    return c;
}


/* 3x3 transform, tracer.cpp:1447-1479 order; matrix rows come from the cold part */
template <typename SP>
__device__ __forceinline__ V3 xform(SP p, int has_trm, V3 in)
{
    float x4 = p->tci[0] * in.x;
    float x5 = p->tcj[1] * in.y;
    float x6 = p->tck[2] * in.z;
    if (has_trm != 1)
    {
        x4 = x4 + p->tci[1] * in.y;
        x4 = x4 + p->tci[2] * in.z;
        x5 = x5 + p->tcj[0] * in.x;
        x5 = x5 + p->tcj[2] * in.z;
        x6 = x6 + p->tck[0] * in.x;
        x6 = x6 + p->tck[1] * in.y;
    }
    V3 o; o.x = x4; o.y = x5; o.z = x6;
    return o;
}

/* ------------------------------------------------------------------------ */
/* per-lane traversal state                                                  */
/* ------------------------------------------------------------------------ */

struct Ray
{
    V3 org, dir;            /* ctx_ORG, ctx_RAY_X..Z                          */
    float tmin, tmax;       /* ctx_T_MIN, initial ctx_T_BUF                   */
    int list;               /* list head element                              */
    int osi;                /* ctx_PARAM(OBJ): originating surface or -1      */
    int oflg;               /* ctx_PARAM(FLG) & 3: side | pass-thru           */
    V3 ploc;                /* parent's local hit (parent ctx_NRM_I..K)       */
};

struct Hit
{
    float t;                /* final ctx_T_BUF                                */
    int si;                 /* surface index or -1                            */
    int side;
    V3 loc;                 /* local (possibly conic-adjusted) hit, ctx_NEW   */
};

/* what a lane carries from one list element to the next (kept minimal: every
 * loop-carried value costs a copy per element) */
struct Walk
{
    V3 dxyz, dijk;          /* ctx_DFF_X..Z / I..K                            */
    V3 rijk;                /* ctx_RAY_I..K                                   */
    float tbuf;             /* ctx_T_BUF                                      */
    int local_obj;          /* ctx_LOCAL(OBJ): trnode's last element          */
    int resume;             /* element at which a bounding-volume skip ends   */
};

/* the hot 80 bytes of a DSurf, fetched with five 16-byte scalar loads issued
 * back to back (one wait) */
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct Hot
{
    float pos0, pos1, pos2; u32 flags;
    float sci0, sci1, sci2, sci3;
    float scj0, scj1, scj2; int clip;
    float min0, min1, min2, d_eps;
    float max0, max1, max2, t_eps;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ Hot ld_hot5(SrfP p)
{
    const QR_CONST u32x4 *q = (const QR_CONST u32x4 *)p;
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    Hot h;
    h.pos0 = u2f(a.x); h.pos1 = u2f(a.y); h.pos2 = u2f(a.z); h.flags = a.w;
    h.sci0 = u2f(b.x); h.sci1 = u2f(b.y); h.sci2 = u2f(b.z); h.sci3 = u2f(b.w);
    h.scj0 = u2f(c.x); h.scj1 = u2f(c.y); h.scj2 = u2f(c.z); h.clip = (int)c.w;
    h.min0 = u2f(d.x); h.min1 = u2f(d.y); h.min2 = u2f(d.z); h.d_eps = u2f(d.w);
    h.max0 = u2f(e.x); h.max1 = u2f(e.y); h.max2 = u2f(e.z); h.t_eps = u2f(e.w);
    return h;
}
/* the same for a per-lane surface (divergent walk): five 16-byte vector loads */
__device__ __forceinline__ Hot ld_hot5(const DSurf *p)
{
    const u32x4 *q = (const u32x4 *)p;
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    Hot h;
    h.pos0 = u2f(a.x); h.pos1 = u2f(a.y); h.pos2 = u2f(a.z); h.flags = a.w;
    h.sci0 = u2f(b.x); h.sci1 = u2f(b.y); h.sci2 = u2f(b.z); h.sci3 = u2f(b.w);
    h.scj0 = u2f(c.x); h.scj1 = u2f(c.y); h.scj2 = u2f(c.z); h.clip = (int)c.w;
    h.min0 = u2f(d.x); h.min1 = u2f(d.y); h.min2 = u2f(d.z); h.d_eps = u2f(d.w);
    h.max0 = u2f(e.x); h.max1 = u2f(e.y); h.max2 = u2f(e.z); h.t_eps = u2f(e.w);
    return h;
}
#pragma clang diagnostic pop

/* sci[axis] for a wave-uniform axis: scalar bit-select (a ternary chain over the
 * struct members is turned into a scratch lookup table by the compiler, which
 * costs a scratch store per list element) */
__device__ __forceinline__ float hsci(const Hot &s, int i)
{
    const u32 m0 = i == 0 ? 0xFFFFFFFFu : 0u, m1 = i == 1 ? 0xFFFFFFFFu : 0u, m2 = i == 2 ? 0xFFFFFFFFu : 0u;
    return u2f((f2u(s.sci0) & m0) | (f2u(s.sci1) & m1) | (f2u(s.sci2) & m2));
}

#include "qr_walk.hpp"
#include "qr_shade.hpp"

/* ------------------------------------------------------------------------ */
/* the kernel                                                                */
/* ------------------------------------------------------------------------ */

__device__ __forceinline__ float clamp1(float x) { return x < 1.0f ? x : 1.0f; }

/* one wave = one schedule entry: footprint `ord`, its tile-list head, rendered into `frame` */
template <bool COUNT, bool DIV>
__device__ __forceinline__ void render_wave(const DevScene &sc, const u32 ord, const int sched_head, const int gw,
                                            uint32_t *__restrict__ frame, int32_t *__restrict__ ids,
                                            unsigned long long *__restrict__ counters)
{
#ifdef QR_WAVETIME
    const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_mid = 0; u32 wt_push = 0;
#endif
    const FrmP fr = c_frm(sc);
    const int fsaa = fr->fsaa;
    const int ns = 1 << fsaa;
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    const int lane = tid & 63;
    const int pix = lane >> fsaa;               /* pixel index inside the wave */
    const int k = lane & (ns - 1);              /* sample index inside the pixel */

    /* wave footprint: 8x8 pixels (no AA), 8x4 (2x), 4x4 (4x).  Every WAVE takes one entry of
     * the host-computed schedule (footprints that can spawn deep recursion first, so that their
     * long waves overlap the bulk instead of forming a tail); consecutive entries are
     * neighbouring footprints, so waves running side by side still share tile lists in the scalar cache. */
    const int fw = fsaa == 2 ? 4 : 8, fh = fsaa == 0 ? 8 : 4;
    (void)wv; (void)gw;
    /* footprints that can recurse get issue priority: the frame ends with the slowest of them, and while
     * the bulk is in flight they would otherwise share their SIMD's issue slots evenly */
    if (ord >> 30) { if ((ord >> 30) >= 2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
    const int px = fsaa == 2 ? (pix & 3) : (pix & 7), py = fsaa == 2 ? (pix >> 2) : (pix >> 3);
    const int x = (int)(ord & 0x3FFFu) * fw + px;
    const int y = (int)((ord >> 14) & 0x3FFFu) * fh + py;
    const int group = y >> 3;

    bool inside = x < fr->frm_w && y < fr->frm_h && y >= sc.row_begin && y < sc.row_end;
    if (group < sc.group_first) inside = false;
    if (sc.group_stride != 1 && (group - sc.group_first) % sc.group_stride != 0) inside = false;
    if (inside && sc.thnum > 1) inside = (y % sc.thnum) == sc.index;
    if (!__any(inside)) return;                 /* whole wave outside this launch's rows */
    if (sched_head == QR_NULL)
    {
        /* empty tile: no ray of the footprint meets anything; the reference's pipeline ends with
         * colour 0 for such a packet (clamp, sqrt and cvt of 0 are 0), so store it and leave */
        if (inside && k == 0)
        {
            frame[(size_t)y * fr->frm_w + x] = 0u;
            if (ids != nullptr) ids[(size_t)y * fr->frm_w + x] = -1;
        }
        if (COUNT)
        {
            unsigned long long n = inside ? 1ull : 0ull;
            for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
            if (lane == 0 && n != 0) atomicAdd(&counters[0], n);
        }
        return;
    }

    Counters cnt = {0, 0, 0, 0};

    /* primary ray, tracer.cpp:1287-1322; sample offsets engine.cpp:3480-3550 */
    Ray ray;
    {
        int ai = 0;
        if (fsaa == 1) ai = (x & 1) * 2 + k;
        if (fsaa == 2) ai = k;
        float ha, va;
        if (fsaa == 0) { ha = fr->hor_a[0]; va = fr->ver_a[0]; }       /* wave-uniform: scalar loads */
        else { ha = ((const float *)sc.frp->hor_a)[ai]; va = ((const float *)sc.frp->ver_a)[ai]; }
        float hs = (float)x + ha; hs = hs + 0.0f;
        float vs = (float)y + va; vs = vs + 0.0f;
        float x1 = fr->hor[0] * hs, x2 = fr->hor[1] * hs, x3 = fr->hor[2] * hs;
        float x4 = fr->ver[0] * vs, x5 = fr->ver[1] * vs, x6 = fr->ver[2] * vs;
        x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
        ray.dir.x = x1 + fr->dir[0];
        ray.dir.y = x2 + fr->dir[1];
        ray.dir.z = x3 + fr->dir[2];
        ray.org.x = fr->org[0]; ray.org.y = fr->org[1]; ray.org.z = fr->org[2];
        ray.tmin = fr->t_min; ray.tmax = fr->t_max;
        ray.osi = QR_NULL; ray.oflg = 0;
        ray.ploc = {0, 0, 0};
        ray.list = QR_NULL;
        if (sched_head != QR_PER_LANE_TILE)
        {
            if (inside) ray.list = sched_head;
        }
        else if (inside)
        {
            const int tile = (y / fr->tile_h) * fr->tls_row + (x / fr->tile_w);
            ray.list = sc.tiles[tile];
        }
    }

    Frame stk[QR_MAX_DEPTH];
#ifdef QR_STATS
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk_trav = 0, tk_shade = 0, tk_rest = 0, tk1;
#define QR_TICK(acc) do { tk1 = __builtin_amdgcn_s_memtime(); acc += tk1 - tk0; tk0 = tk1; } while (0)
#else
#define QR_TICK(acc) do { } while (0)
#endif
    Outer ou;
    ou.sp = 0;
    ou.mode = inside ? 0 : 2;                   /* 0 trace, 1 return, 2 done */
    ou.ret = {0, 0, 0};
    ou.hit_id = -1;
    int &sp = ou.sp, &mode = ou.mode, &hit_id = ou.hit_id;
    V3 &ret = ou.ret;
    const int depth = sc.depth;

    if (COUNT && inside) cnt.primary++;

    while (__any(mode != 2))
    {
        const bool tr = mode == 0;
        if (__any(tr))
        {
            Hit h; bool occ;
            QR_TICK(tk_rest);
#ifdef QR_WAVETIME
            const unsigned long long wt_a = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef QR_X_NOTRACE
            h.t = ray.tmax; h.si = ray.list; h.side = 0; h.loc = ray.ploc; occ = false;
#else
            traverse<false, DIV>(sc, tr, ray, h, occ);
#endif
#ifdef QR_WAVETIME
            const unsigned long long wt_b = __builtin_amdgcn_s_memrealtime();
#endif
            QR_TICK(tk_trav);
#ifdef QR_WAVETIME
            if (wt_mid == 0) wt_mid = __builtin_amdgcn_s_memrealtime();
            wt_push++;
#endif
            const bool got = tr && h.si != QR_NULL && !QR_KNOB(4);
            if (tr && !got) { ret = {0, 0, 0}; mode = 1; }
            if (got && sp == 0) hit_id = (h.si << 1) | h.side;

            Shaded o;
            shade<COUNT, DIV>(sc, got, ray, h, o, cnt);
#ifdef QR_WAVETIME
            if (__ffsll((long long)__ballot(true)) - 1 == lane)
            {
                unsigned long long *o_ = sc.stats + 28 + (size_t)gw * QR_WT_SLOTS;
                o_[8] += wt_b - wt_a; o_[9] += __builtin_amdgcn_s_memrealtime() - wt_b;
            }
#endif
            QR_TICK(tk_shade);

            if (got)
            {
                const bool can_spawn = (depth - sp) != 0;
                const int meta = (h.si << 4) | (h.side << 3) | (o.want_rf ? 4 : 0);
                if (o.want_tr && can_spawn)
                {
                    Frame &f = stk[sp];
                    f.col[0] = o.col.x; f.col[1] = o.col.y; f.col[2] = o.col.z;
                    f.c_trn = o.c_trn; f.c_rfl = o.c_rfl; f.x0 = o.x0;
                    f.rdir[0] = o.rdir.x; f.rdir[1] = o.rdir.y; f.rdir[2] = o.rdir.z;
                    f.hit[0] = o.hit.x; f.hit[1] = o.hit.y; f.hit[2] = o.hit.z;
                    f.loc[0] = o.loc.x; f.loc[1] = o.loc.y; f.loc[2] = o.loc.z;
                    f.meta = meta | 1;
                    sp++;
                    ray.org = o.hit; ray.dir = o.tdir; ray.tmin = 0.0f; ray.tmax = fr->t_max;
                    ray.list = o.lst_tr; ray.osi = h.si; ray.oflg = h.side | FLAG_PASS_THRU;
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
                        Frame &f = stk[sp];
                        f.col[0] = c.x; f.col[1] = c.y; f.col[2] = c.z;
                        f.c_trn = o.c_trn; f.c_rfl = o.c_rfl; f.x0 = o.x0;
                        f.rdir[0] = o.rdir.x; f.rdir[1] = o.rdir.y; f.rdir[2] = o.rdir.z;
                        f.hit[0] = o.hit.x; f.hit[1] = o.hit.y; f.hit[2] = o.hit.z;
                        f.loc[0] = o.loc.x; f.loc[1] = o.loc.y; f.loc[2] = o.loc.z;
                        f.meta = meta | 2;
                        sp++;
                        ray.org = o.hit; ray.dir = o.rdir; ray.tmin = 0.0f; ray.tmax = fr->t_max;
                        ray.list = o.lst_rf; ray.osi = h.si; ray.oflg = h.side;
                        ray.ploc = o.loc;
                        mode = 0;
                        if (COUNT) cnt.reflect++;
                    }
                    else
                    {
                        if (o.want_rf) { c.x = 0.0f + c.x; c.y = 0.0f + c.y; c.z = 0.0f + c.z; }
                        ret = c;
                        mode = 1;
                    }
                }
            }
        }

        if (mode == 1)
        {
            if (sp == 0)
            {
                mode = 2;
            }
            else
            {
                Frame &f = stk[sp - 1];
                const int phase = f.meta & 3;
                if (phase == 1)
                {
                    /* TR_ret + TR_mix 3534-3598 */
                    V3 c;
                    c.x = ret.x * f.c_trn + f.col[0] * f.x0;
                    c.y = ret.y * f.c_trn + f.col[1] * f.x0;
                    c.z = ret.z * f.c_trn + f.col[2] * f.x0;
                    if (f.meta & 4)
                    {
                        /* reflection child of the same node (depth budget is
                         * the same as for the refraction child) */
                        f.col[0] = c.x; f.col[1] = c.y; f.col[2] = c.z;
                        f.meta = (f.meta & ~3) | 2;
                        const int psi = f.meta >> 4, pside = (f.meta >> 3) & 1;
                        ray.org = {f.hit[0], f.hit[1], f.hit[2]};
                        ray.dir = {f.rdir[0], f.rdir[1], f.rdir[2]};
                        ray.tmin = 0.0f; ray.tmax = fr->t_max;
                        ray.list = sc.shd[psi].lst[pside * 2 + 1];
                        ray.osi = psi; ray.oflg = pside;
                        ray.ploc = {f.loc[0], f.loc[1], f.loc[2]};
                        mode = 0;
                        if (COUNT) cnt.reflect++;
                    }
                    else
                    {
                        ret = c;
                        sp--;
                    }
                }
                else
                {
                    /* RF_ret + RF_mix 3868-3908 */
                    ret.x = ret.x * f.c_rfl + f.col[0];
                    ret.y = ret.y * f.c_rfl + f.col[1];
                    ret.z = ret.z * f.c_rfl + f.col[2];
                    sp--;
                }
            }
        }
    }

    QR_TICK(tk_rest);
#ifdef QR_STATS
    if (lane == 0)
    {
        atomicAdd(&sc.stats[9], tk_trav); atomicAdd(&sc.stats[10], tk_shade); atomicAdd(&sc.stats[11], tk_rest);
    }
#endif
#ifdef QR_WAVETIME
    if (!COUNT && __ffsll((long long)__ballot(true)) - 1 == lane)
    {
        unsigned long long *o = counters + 32 + (size_t)gw * QR_WT_SLOTS;
        o[0] = wt_start; o[1] = wt_mid; o[2] = __builtin_amdgcn_s_memrealtime();
        o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))
             | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 32)
             | ((unsigned long long)wt_push << 40);
    }
#endif
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
    if (inside && k == 0)
    {
        if (fr->ctx_flags & QR_PROP_GAMMA)
        {
            asm volatile("" ::: "memory");      /* keep the branch: three IEEE square roots are not worth speculating */
            cr = __builtin_sqrtf(cr); cg = __builtin_sqrtf(cg); cb = __builtin_sqrtf(cb);
        }
        cr = cr * fr->clamp; cg = cg * fr->clamp; cb = cb * fr->clamp;
        const u32 p = (((u32)cvt_near(cr) & fr->cmask) << 16) |
                      (((u32)cvt_near(cg) & fr->cmask) << 8) |
                       ((u32)cvt_near(cb) & fr->cmask);
        frame[(size_t)y * fr->frm_w + x] = p;
        if (ids != nullptr) ids[(size_t)y * fr->frm_w + x] = hit_id;
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

/* single-scene launch: the scene record travels in the kernel arguments */
template <bool COUNT, int WAVES, bool DIV = false>
__global__ __launch_bounds__(QR_BLOCK, WAVES)
void qr_render_kernel(DevScene sc, uint32_t *__restrict__ frame, int32_t *__restrict__ ids,
                      unsigned long long *__restrict__ counters)
{
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (QR_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (gw >= sc.n_blocks) return;
    /* schedule entry {footprint coordinates, tile-list head or QR_PER_LANE_TILE}: one scalar load */
    typedef u32 u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ sched = ((const QR_CONST u32x2_ *)sc.order)[gw];
    render_wave<COUNT, DIV>(sc, sched.x, (int)sched.y, gw, frame, ids, counters);
}

/*
 * Multi-target launch (qr_render_multi_async): ONE grid renders row ranges of several frames -- of the
 * same or of different scenes -- so that the N blocks a GPU owns in a sharded step share one launch: a
 * frame cut into N small launches pays N ramps, drains and tails (0.23 ms for 8 blocks of demo1 at
 * 1080p against 0.09 ms for the whole frame).  Schedule entries are 16 bytes {footprint, tile-list
 * head, target index, 0}, heavy footprints of all targets first.
 */
#define QR_MAX_TARGETS 16
struct DevTarget { uint32_t *frame; int32_t row_begin, row_end; int32_t scene, pad; };
struct DevTargets { DevTarget t[QR_MAX_TARGETS]; };

template <int WAVES>
__global__ __launch_bounds__(QR_BLOCK, WAVES)
void qr_render_multi_kernel(const DevScene *__restrict__ scenes, DevTargets tg,
                            const uint32_t *__restrict__ order16, int n_blocks,
                            unsigned long long *__restrict__ counters)
{
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (QR_BLOCK / 64) + (int)(threadIdx.x >> 6));
    if (gw >= n_blocks) return;
    const u32x4 sched = ((const QR_CONST u32x4 *)order16)[gw];
    const int ti = (int)sched.z;
    const DevTarget t = tg.t[ti];
    DevScene sc;
    {
        /* dword-wise copy from the constant address space (scalar loads) */
        const QR_CONST u32 *src = (const QR_CONST u32 *)(scenes + t.scene);
        u32 *dst = (u32 *)&sc;
#pragma unroll
        for (unsigned i = 0; i < sizeof(DevScene) / 4; i++) dst[i] = src[i];
    }
    sc.row_begin = t.row_begin; sc.row_end = t.row_end;
    sc.index = 0; sc.thnum = 1;
    sc.group_first = t.row_begin / 8; sc.group_stride = 1;
    render_wave<false, false>(sc, sched.x, (int)sched.y, gw, t.frame, nullptr, counters);
}

#endif /* QR_KERNEL_HPP */
