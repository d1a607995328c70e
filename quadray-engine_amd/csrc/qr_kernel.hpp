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

/* ------------------------------------------------------------------------ */
#ifdef QR_STATS2
#define QR_TT(x) x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#endif
/* CC_clp, tracer.cpp:1597-2160.  `s`, `P` and the clipper list are          */
/* wave-uniform; every temporary is local to the call.  `loc` returns the    */
/* local hit (ctx_NEW_* of the surface's space).                             */
/* ------------------------------------------------------------------------ */

template <bool DIV, typename SP>
__device__ __forceinline__ u32 clip(const DevScene &sc, const Hot &s, SP P,
                                     const Ray &r, const Walk &w, const V3 &df,
                                     bool dmask, u32 amask, float t, int side, u32 m, V3 &loc
#ifdef QR_STATS2
                                     , unsigned long long *g_clip
#endif
                                     )
{
#ifdef QR_STATS2
    unsigned long long g_c; g_c = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    const u32 fl = s.flags;
    const int has_trm = (int)DF_TRM(fl);
    float x4, x5, x6;
    V3 hit;

    m &= LM(cgt(w.tbuf, t)) & LM(clt(r.tmin, t));

    x4 = r.dir.x * t; x4 = x4 + r.org.x; hit.x = x4;
    x5 = r.dir.y * t; x5 = x5 + r.org.y; hit.y = x5;
    x6 = r.dir.z * t; x6 = x6 + r.org.z; hit.z = x6;

    V3 nijk = {0.0f, 0.0f, 0.0f};       /* ctx_NEW_I..K, only defined when has_trm != 0 */
    if (has_trm != 0)
    {
        x4 = w.rijk.x * t; x4 = x4 + w.dijk.x;
        x5 = w.rijk.y * t; x5 = x5 + w.dijk.y;
        x6 = w.rijk.z * t; x6 = x6 + w.dijk.z;
        nijk.x = x4; nijk.y = x5; nijk.z = x6;
    }
    else
    {
        x4 = x4 - s.pos0;
        x5 = x5 - s.pos1;
        x6 = x6 - s.pos2;
    }
    /* the local hit the surface's own solvers/material see: NEW[shift] */
    const int sh = (int)DF_SHIFT(fl);
    V3 nw;
    if ((sh != 0) == (has_trm != 0)) { nw.x = x4; nw.y = x5; nw.z = x6; }
    else if (sh) { nw = nijk; }                     /* shift without transform: stale IJK (never built by the engine) */
    else { nw.x = hit.x - s.pos0; nw.y = hit.y - s.pos1; nw.z = hit.z - s.pos2; }

    /* conic singularity solver, 1706-1856 */
    const int conic = (int)DF_CONIC(fl);
    if (conic != 0)
    {
        const int mi = (int)DF_MAP(fl, 0), mj = (int)DF_MAP(fl, 1), mk = (int)DF_MAP(fl, 2);
        float x0, x1, x2, x3;
        x1 = vget(nw, mi); x1 = x1 * x1; x0 = x1;
        if (conic != 2) { x2 = vget(nw, mj); x2 = x2 * x2; x0 = x0 + x2; }
        x3 = vget(nw, mk); x3 = x3 * x3; x0 = x0 + x3;
        const bool hm = clt(x0, s.t_eps) && dmask;
        if (hm)
        {
            const u32 sm = QR_SMASK;
            const float one = 1.0f;
            float r4;
            x2 = 0.0f;
            x1 = u2f((f2u(vget(df, mi)) & sm) ^ f2u(one));
            x3 = hsci(s, mi);
            r4 = one;
            if (conic != 2)
            {
                x2 = u2f((f2u(vget(df, mj)) & sm) ^ f2u(one));
                x3 = x3 + hsci(s, mj);
                r4 = r4 + one;
            }
            x3 = x3 / hsci(s, mk);
            x3 = fxor(x3, sm);
            float y6 = x3;
            x3 = __builtin_sqrtf(x3);
            y6 = y6 + r4;
            r4 = rsq(y6);
            r4 = r4 * s.t_eps;
            x1 = x1 * r4; x2 = x2 * r4; x3 = x3 * r4;

            const u32 tside = side ? sm : 0u;
            x3 = fxor(x3, f2u(vget(df, mk)) & sm);
            x3 = fxor(x3, (tside & amask) ^ amask);
            const u32 u5 = (tside | amask) ^ amask;
            x1 = fxor(x1, u5);
            x2 = fxor(x2, u5);

            vset(nw, mi, x1);
            if (conic != 2) vset(nw, mj, x2);
            vset(nw, mk, x3);
            if (sh) nijk = nw;
            x4 = nw.x; x5 = nw.y; x6 = nw.z;
        }
    }
    loc = nw;

    /* axis min/max, 1874-1927: the upload replaces the bound of an unclipped axis by -inf/+inf,
     * which makes the six compares unconditional (a lane still in `m` has a finite hit point) */
    m &= LM(cle(s.min0, x4)) & LM(cge(s.max0, x4));
    m &= LM(cle(s.min1, x5)) & LM(cge(s.max1, x5));
    m &= LM(cle(s.min2, x6)) & LM(cge(s.max2, x6));

#ifdef QR_STATS2
    { unsigned long long t_; QR_TT(t_); g_clip[0] += t_ - g_c; g_c = t_; g_clip[2] += 1; }
#endif
    /* custom clipping, 1931-2151 */
    int e = s.clip;
    if (DIV)
    {
        /* the same loop with a per-lane clipper list: every lane steps through its own cells (vector loads),
         * `continue` of the wave-uniform version becomes `break` out of the one-trip do-block */
        int redx = QR_NULL;
        const int local_lst = P->trnode;
        u32 c_acc = 0;
        V3 cxyz = {0.0f, 0.0f, 0.0f}, cijk = {0.0f, 0.0f, 0.0f};
        if (!__any(m != 0)) e = QR_NULL;
        while (__any(e != QR_NULL))
        {
            if (e != QR_NULL)
            do
            {
                const DCell dc_ = sc.elm[e];
                qr_elem el; el.simd = dc_.simd; el.data = dc_.data; el.next = dc_.next; el.kind = dc_.kind;
                const int ecur = e;
                e = el.next;
                if (el.simd == QR_NULL)
                {
                    if (el.data > 0) { m = ~m & c_acc; }
                    else             { c_acc = m; m = DF_CDEF(fl) != 0 ? 0xFFFFFFFFu : 0u; }
                    break;
                }
                const DSurf *kp = sc.srf + el.simd;
                const Hot k = ld_hot5(kp);
                const u32 kf = k.flags;
                const int ktrm = (int)DF_TRM(kf);
                const bool karr = DF_ARRAY(kf) != 0;
                bool have_vec = false;
                if (!karr)
                {
                    if (redx != QR_NULL)
                    {
                        cijk.x = cxyz.x - k.pos0;
                        cijk.y = cxyz.y - k.pos1;
                        cijk.z = cxyz.z - k.pos2;
                        if (ecur == redx) redx = QR_NULL;
                        have_vec = true;
                    }
                }
                else if (el.simd == local_lst)
                {
                    cxyz.x = nijk.x + s.pos0;
                    cxyz.y = nijk.y + s.pos1;
                    cxyz.z = nijk.z + s.pos2;
                    redx = el.data;
                    break;
                }
                if (!have_vec)
                {
                    V3 d;
                    d.x = hit.x - k.pos0;
                    d.y = hit.y - k.pos1;
                    d.z = hit.z - k.pos2;
                    cxyz = d;
                    if (ktrm != 0)
                    {
                        V3 p = xform(kp, ktrm, d);
                        if (karr)
                        {
                            cxyz = p;
                            redx = el.data;
                            break;
                        }
                        cijk = p;
                    }
                }
                const V3 cv = DF_SHIFT(kf) ? cijk : cxyz;
                const int ckind = (int)DF_CKIND(kf);
                float f4 = 0.0f, f5, f6, f1, f2, f3;
                bool ok = true;
                if (ckind == 1)
                {
                    f4 = fxor(vget(cv, (int)DF_MAP(kf, 2)), DF_SGN(kf, 2));
                }
                else if (ckind == 2)
                {
                    f4 = cv.x; f1 = k.scj0; f1 = f1 + f1; f1 = f1 * f4;
                    f4 = f4 * f4; f4 = f4 * k.sci0; f4 = f4 - f1;
                    f5 = cv.y; f2 = k.scj1; f2 = f2 + f2; f2 = f2 * f5;
                    f5 = f5 * f5; f5 = f5 * k.sci1; f5 = f5 - f2;
                    f6 = cv.z; f3 = k.scj2; f3 = f3 + f3; f3 = f3 * f6;
                    f6 = f6 * f6; f6 = f6 * k.sci2; f6 = f6 - f3;
                    f4 = f4 - k.sci3; f4 = f4 + f5; f4 = f4 + f6;
                }
                else if (ckind == 3)
                {
                    f4 = cv.x; f4 = f4 * f4; f4 = f4 * k.sci0;
                    f5 = cv.y; f5 = f5 * f5; f5 = f5 * k.sci1;
                    f6 = cv.z; f6 = f6 * f6; f6 = f6 * k.sci2;
                    f4 = f4 - k.sci3; f4 = f4 + f5; f4 = f4 + f6;
                }
                else
                {
                    ok = false;
                }
                if (ok) m &= LM(el.data < 0 ? cge(f4, 0.0f) : cle(f4, 0.0f));
            }
            while (0);
        }
        e = QR_NULL;
    }
    if (e != QR_NULL && __any(m != 0))
    {
        int redx = QR_NULL;
        const int local_lst = P->trnode;
        u32 c_acc = 0;
        V3 cxyz = {0.0f, 0.0f, 0.0f}, cijk = {0.0f, 0.0f, 0.0f};   /* ctx_NRM_* as clip temporaries */
        while (e != QR_NULL)
        {
            e = __builtin_amdgcn_readfirstlane(e);
            const qr_elem el = ld_elem(c_elm(sc) + e);
            const int enext = el.next;
            if (el.simd == QR_NULL)
            {
                if (el.data > 0) { m = ~m & c_acc; }
                else             { c_acc = m; m = DF_CDEF(fl) != 0 ? 0xFFFFFFFFu : 0u; }
                e = enext;
                continue;
            }
            SrfP kp = c_srf(sc) + el.simd;
            const Hot k = ld_hot5(kp);
            const u32 kf = k.flags;
            const int ktrm = (int)DF_TRM(kf);
            const bool karr = DF_ARRAY(kf) != 0;
            bool have_vec = false;
            if (!karr)
            {
                if (redx != QR_NULL)
                {
                    cijk.x = cxyz.x - k.pos0;
                    cijk.y = cxyz.y - k.pos1;
                    cijk.z = cxyz.z - k.pos2;
                    if (e == redx) redx = QR_NULL;
                    have_vec = true;
                }
            }
            else if (el.simd == local_lst)
            {
                cxyz.x = nijk.x + s.pos0;
                cxyz.y = nijk.y + s.pos1;
                cxyz.z = nijk.z + s.pos2;
                redx = el.data;
                e = enext;
                continue;
            }
            if (!have_vec)
            {
                V3 d;
                d.x = hit.x - k.pos0;
                d.y = hit.y - k.pos1;
                d.z = hit.z - k.pos2;
                cxyz = d;
                if (ktrm != 0)
                {
                    V3 p = xform(kp, ktrm, d);
                    if (karr)
                    {
                        cxyz = p;
                        redx = el.data;
                        e = enext;
                        continue;
                    }
                    cijk = p;
                }
            }
            {
                const V3 cv = DF_SHIFT(kf) ? cijk : cxyz;
                const int ckind = (int)DF_CKIND(kf);
                float f4 = 0.0f, f5, f6, f1, f2, f3;
                bool ok = true;
                if (ckind == 1)
                {
                    f4 = fxor(vget(cv, (int)DF_MAP(kf, 2)), DF_SGN(kf, 2));
                }
                else if (ckind == 2)
                {
                    f4 = cv.x; f1 = k.scj0; f1 = f1 + f1; f1 = f1 * f4;
                    f4 = f4 * f4; f4 = f4 * k.sci0; f4 = f4 - f1;
                    f5 = cv.y; f2 = k.scj1; f2 = f2 + f2; f2 = f2 * f5;
                    f5 = f5 * f5; f5 = f5 * k.sci1; f5 = f5 - f2;
                    f6 = cv.z; f3 = k.scj2; f3 = f3 + f3; f3 = f3 * f6;
                    f6 = f6 * f6; f6 = f6 * k.sci2; f6 = f6 - f3;
                    f4 = f4 - k.sci3; f4 = f4 + f5; f4 = f4 + f6;
                }
                else if (ckind == 3)
                {
                    f4 = cv.x; f4 = f4 * f4; f4 = f4 * k.sci0;
                    f5 = cv.y; f5 = f5 * f5; f5 = f5 * k.sci1;
                    f6 = cv.z; f6 = f6 * f6; f6 = f6 * k.sci2;
                    f4 = f4 - k.sci3; f4 = f4 + f5; f4 = f4 + f6;
                }
                else
                {
                    ok = false;
                }
                if (ok)
                {
                    m &= LM(el.data < 0 ? cge(f4, 0.0f) : cle(f4, 0.0f));
                }
            }
            e = enext;
        }
    }
#ifdef QR_STATS2
    { unsigned long long t_; QR_TT(t_); g_clip[1] += t_ - g_c; }
#endif
    return m;
}

/* ------------------------------------------------------------------------ */
/* one list element for the lanes of a group (everything about the element   */
/* and its surface is wave-uniform): tracer.cpp:1341-1592, 3955-4054,        */
/* 4062-4136, 4216-4277, 4378-4842                                           */
/* ------------------------------------------------------------------------ */

template <bool SHADOW, bool DIV, typename SP>
__device__ __forceinline__ int walk_element(const DevScene &sc, const int e, const qr_elem &el, SP P,
                                            const Ray &r, Walk &w, Hit &h, bool &occluded, bool &live
#ifdef QR_STATS2
                                             , unsigned long long *g_seg
#endif
                                             )
{
#ifdef QR_STATS2
    unsigned long long g_t; QR_TT(g_t);
#endif
    const bool on = live && w.resume == QR_NULL;

    if (__any(on))
    {
        const Hot s = ld_hot5(P);
#ifdef QR_STATS2
        asm volatile("" :: "s"(s.flags), "s"(s.max2));
        { unsigned long long t_; QR_TT(t_); g_seg[0] += t_ - g_t; g_t = t_; }
#endif
        const int si = el.simd;
        const u32 fl = s.flags;
        const bool is_arr = DF_ARRAY(fl) != 0;
        const int has_trm = (int)DF_TRM(fl);
        const int sh = (int)DF_SHIFT(fl);

        if (on)
        {
            const bool same = si == r.osi;

            /* ---- diff / ray in the surface's space, 1352-1556 ---- */
            if (same)
            {
                if (sh) w.dijk = r.ploc; else w.dxyz = r.ploc;
            }
            if (!is_arr && w.local_obj != QR_NULL)
            {
                if (!same)
                {
                    w.dijk.x = w.dxyz.x - s.pos0;
                    w.dijk.y = w.dxyz.y - s.pos1;
                    w.dijk.z = w.dxyz.z - s.pos2;
                }
                if (e == w.local_obj) w.local_obj = QR_NULL;
            }
            else
            {
                bool do_ray = true;
                if (!same)
                {
                    V3 d;
                    d.x = r.org.x - s.pos0;
                    d.y = r.org.y - s.pos1;
                    d.z = r.org.z - s.pos2;
                    w.dxyz = d;
                    if (has_trm == 0)
                    {
                        do_ray = false;
                    }
                    else
                    {
                        V3 p = xform(P, has_trm, d);
                        if (is_arr) { w.dxyz = p; w.local_obj = el.data; }
                        else        { w.dijk = p; }
                    }
                }
                if (do_ray) w.rijk = xform(P, has_trm, r.dir);
            }

#ifdef QR_STATS2
            { unsigned long long t_; QR_TT(t_); g_seg[1] += t_ - g_t; g_t = t_; }
#endif
            const V3 ry = sh ? w.rijk : r.dir;
            const V3 df = sh ? w.dijk : w.dxyz;

            if ((el.kind & 3) == 1)
            {
                /* AR_ptr 3955-4054 */
                float x0, x1, x2, x3, x4, x5, x6, x7;
                x1 = ry.x; x0 = s.sci0 * x1; x5 = df.x; x7 = s.sci0 * x5;
                x3 = x1; x1 = x1 * x0; x3 = x3 * x7; x5 = x5 * x7;
                x2 = ry.y; x0 = s.sci1 * x2; x6 = df.y; x7 = s.sci1 * x6;
                x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
                x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
                x2 = ry.z; x0 = s.sci2 * x2; x6 = df.z; x7 = s.sci2 * x6;
                x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
                x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
                x5 = x5 - s.sci3;
                x5 = x5 * x1;
                x3 = x3 * x3;
                x3 = x3 - x5;
                if (!cle(0.0f, x3))
                {
                    w.resume = el.data;
                    if (w.resume == w.local_obj) w.local_obj = QR_NULL;
                }
            }
            else
            {
                const int solver = (SHADOW && QR_KNOB(32)) ? 0 : (int)DF_SOLVER(fl);
                /* up to two candidate roots per lane, in the lane's own order */
                float ct0 = 0.0f, ct1 = 0.0f;
                int   cs0 = 0, cs1 = 0;
                bool  cm0 = false, cm1 = false;
                int   ncand = 0;
                bool  dmask = false;
                u32   amask = 0;

                if (solver == 1)
                {
                    /* PL_ptr 4062-4136 */
                    const int mk = (int)DF_MAP(fl, 2);
                    const u32 sg = DF_SGN(fl, 2);
                    float dk = fxor(vget(df, mk), sg);
                    const float rk = fxor(vget(ry, mk), sg);
                    dk = fxor(dk, QR_SMASK);
                    cm0 = !same && cne(0.0f, rk);
                    /* Pre-test (ours): the hit only survives clip() if t_min < t < t_buf.  With t_min >= 0 a
                     * quotient of opposite signs cannot, and |dk| >= |rk| * t_buf * (1 + 2^-20) means
                     * t >= t_buf whatever the rounding of the division; dropping those lanes here changes
                     * nothing, and when no lane is left the wave skips the IEEE division and clip(). */
                    {
                        const bool opposite = ((f2u(dk) ^ f2u(rk)) & QR_SMASK) != 0;
                        const bool beyond = fabs_bits(dk) >= fabs_bits(rk) * (w.tbuf * 1.000001f);
                        cm0 = cm0 && !((opposite || beyond) && r.tmin >= 0.0f);
                    }
                    if (__any(cm0)) ct0 = dk / rk;
                    cs0 = clt(rk, 0.0f) ? 0 : 1;
                    ncand = 1;
                }
                else if (solver != 0)
                {
                    float a, b, c, d;
                    if (solver == 2)
                    {
                        /* QD_ptr 4378-4447 */
                        float x0, x1, x2, x3, x4, x5, x6, x7;
                        x1 = ry.x; x0 = s.sci0 * x1; x5 = df.x; x7 = s.sci0 * x5;
                        x7 = x7 - s.scj0; x3 = x1; x1 = x1 * x0; x3 = x3 * x7; x7 = x7 - s.scj0; x5 = x5 * x7;
                        x2 = ry.y; x0 = s.sci1 * x2; x6 = df.y; x7 = s.sci1 * x6;
                        x7 = x7 - s.scj1; x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x7 = x7 - s.scj1; x6 = x6 * x7;
                        x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
                        x2 = ry.z; x0 = s.sci2 * x2; x6 = df.z; x7 = s.sci2 * x6;
                        x7 = x7 - s.scj2; x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x7 = x7 - s.scj2; x6 = x6 * x7;
                        x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
                        x5 = x5 - s.sci3;
                        x6 = x5; x5 = x5 * x1; x4 = x3; x3 = x3 * x3; x3 = x3 - x5;
                        a = x1; b = x4; c = x6; d = x3;
                    }
                    else
                    {
                        /* TP_ptr 4216-4277 */
                        const int mi = (int)DF_MAP(fl, 0), mk = (int)DF_MAP(fl, 2);
                        float x0, x1, x2, x3, x4, x5, x6, x7;
                        x1 = vget(ry, mi); x5 = vget(df, mi); x3 = hsci(s, mi);
                        x2 = vget(ry, mk); x6 = vget(df, mk); x4 = hsci(s, mk);
                        x0 = x5; x7 = x6;
                        x6 = x6 * x1; x5 = x5 * x2; x5 = x5 - x6; x5 = x5 * x5; x5 = x5 * x3; x5 = x5 * x4;
                        x5 = fabs_bits(x5);
                        x6 = x3; x3 = x3 * x0; x4 = x4 * x7; x3 = x3 * x1; x4 = x4 * x2; x3 = x3 + x4;
                        x4 = hsci(s, mk);
                        x0 = x0 * x0; x7 = x7 * x7; x0 = x0 * x6; x7 = x7 * x4; x0 = x0 + x7;
                        x1 = x1 * x1; x2 = x2 * x2; x1 = x1 * x6; x2 = x2 * x4; x1 = x1 + x2;
                        a = x1; b = x3; c = x0; d = x5;
                    }

                    /* QD_rts 4449-4658 */
                    const u32 sm = QR_SMASK;
                    const bool xmask = cle(0.0f, d);
                    /* CHECK_MASK(OO_end, NONE, xmask), 4455 */
                    if (__any(xmask))
                    {
                        b = fxor(b, sm);
                        dmask = xmask && clt(d, s.d_eps);

                        const float sd = fxor(__builtin_sqrtf(d), sm & f2u(b));
                        const float bd = b + sd;
                        const bool m_pos = cle(0.0f, sd);
                        const bool m_neg = cgt(0.0f, sd);
                        const float t2n = u2f((m_neg ? f2u(c) : 0u)  | (m_pos ? f2u(bd) : 0u));
                        const float t1n = u2f((m_neg ? f2u(bd) : 0u) | (m_pos ? f2u(c) : 0u));
                        float t2d = u2f((m_neg ? f2u(bd) : 0u) | (m_pos ? f2u(a) : 0u));
                        float t1d = u2f((m_neg ? f2u(a) : 0u)  | (m_pos ? f2u(bd) : 0u));
                        a = u2f((m_pos ? f2u(a) : 0u) | (m_neg ? f2u(a) : 0u));

                        amask = sm & f2u(a);
                        if (dmask)
                        {
                            if (ceq(t1n, 0.0f)) t1d = 1.0f;
                            if (ceq(t2n, 0.0f)) t2d = 1.0f;
                        }
                        float t1 = t1n / t1d;
                        float t2 = t2n / t2d;
                        const bool t1msk = cne(t1d, 0.0f);
                        const bool t2msk = cne(t2d, 0.0f);
                        if (dmask)
                        {
                            float tdf = t1 - t2;
                            tdf = fxor(tdf, amask);
                            const bool f = cle(0.0f, tdf);
                            tdf = f ? tdf : 0.0f;
                            float eps = f ? s.t_eps : 0.0f;
                            eps = eps * t1;
                            eps = fabs_bits(eps);
                            tdf = tdf * -0.5f;
                            tdf = tdf - eps;
                            tdf = fxor(tdf, amask);
                            tdf = (t1msk && t2msk) ? tdf : 0.0f;
                            t1 = t1 + tdf;
                            t2 = t2 - tdf;
                        }

                        const bool inner_first = xmask && cgt(0.0f, a);
                        /* CHECK_SIDE 531-540 */
                        const int f3 = r.oflg & (FLAG_SIDE | FLAG_PASS_THRU);
                        const bool skip_outer = same && (f3 == 1 || f3 == 2);
                        const bool skip_inner = same && (f3 == 0 || f3 == 3);
                        const bool mo = xmask && t1msk && !skip_outer;
                        const bool mi2 = xmask && t2msk && !skip_inner;
                        ncand = 2;
                        if (inner_first) { ct0 = t2; cs0 = 1; cm0 = mi2; ct1 = t1; cs1 = 0; cm1 = mo; }
                        else             { ct0 = t1; cs0 = 0; cm0 = mo;  ct1 = t2; cs1 = 1; cm1 = mi2; }
                    }
                }

#ifdef QR_STATS2
                { unsigned long long t_; QR_TT(t_); g_seg[2] += t_ - g_t; g_t = t_; }
#endif
                bool done = false;
#pragma nounroll
                for (int p = 0; p < ncand; p++)
                {
                    const float t = p == 0 ? ct0 : ct1;
                    const int side = p == 0 ? cs0 : cs1;
                    u32 m = ((p == 0 ? cm0 : cm1) && !done) ? 0xFFFFFFFFu : 0u;
                    if (!__any(m != 0) || (SHADOW && QR_KNOB(16))) continue;
                    V3 loc;
                    m = clip<DIV>(sc, s, P, r, w, df, dmask, amask, t, side, m, loc
#ifdef QR_STATS2
                             , g_seg + 4
#endif
                             );
                    if (m != 0)
                    {
                        done = true;
                        if (SHADOW)
                        {
                            /* CHECK_SHAD 549-589 */
                            const int props = side ? P->props1 : P->props0;
                            const bool no_shadow = (props & QR_PROP_LIGHT) ||
                                                   ((props & QR_PROP_TRANSP) && !(props & QR_PROP_REFRACT));
                            if (!no_shadow) { occluded = true; live = false; }
                        }
                        else
                        {
                            /* PAINT_FRAG 653-662: depth write; shading is deferred */
                            w.tbuf = t;
                            h.t = t; h.si = si; h.side = side;
                            h.loc = loc;
                        }
                    }
                }
            }
        }
    }

#ifdef QR_STATS2
    { unsigned long long t_; QR_TT(t_); g_seg[3] += t_ - g_t; g_t = t_; }
#endif
    if (w.resume == e) w.resume = QR_NULL;
    /*
     * The reference jumps a whole packet to the end of an array whose bounding volume no lane hits
     * (tracer.cpp:4040-4054); here rays skip individually, so take the jump when this array head
     * left no live ray of the group walking.  Rays that were skipping already wait for the end of an
     * enclosing array, which lies at or behind this array's end when arrays are properly nested
     * (sc.nested, verified at upload); otherwise jump only if all rays wait for this array's end.
     */
    if (DIV)
    {
        /* every ray walks alone: one that missed this array's volume goes straight to the array's end */
        return ((el.kind & 3) == 1 && w.resume == el.data) ? el.data : QR_NULL;
    }
    if ((el.kind & 3) == 1)
    {
        if (sc.nested ? !__any(live && w.resume == QR_NULL) : !__any(live && w.resume != el.data)) return el.data;
    }
    return QR_NULL;
}

/*
 * OO_cyc for a group of lanes that share the list `head` (wave-uniform, not NULL).
 */
template <bool SHADOW>
__device__ __forceinline__ void walk_list(const DevScene &sc, int head, const Ray &r, Hit &h, bool &occluded)
{
    Walk w;
    w.dxyz = {0, 0, 0}; w.dijk = {0, 0, 0}; w.rijk = {0, 0, 0};
    w.tbuf = r.tmax;
    w.local_obj = QR_NULL;
    w.resume = QR_NULL;

    bool live = true;
    const ElmP E = c_elm(sc);
    const SrfP D = c_srf(sc);
    int e = __builtin_amdgcn_readfirstlane(head);
    const float dd = r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z;
    /* only the cull uses the ray length: an upper bound is enough there, so the 1-instruction
     * approximate square root (1 ulp) inflated by 2^-20 replaces the IEEE expansion */
    const float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
#ifdef QR_STATS
    unsigned long long st_iter = 0, st_lanes = 0, st_skip = 0;
#endif
#ifdef QR_STATS2
    unsigned long long tA = 0, tB = 0, tC = 0, nA = 0, nC = 0, t0, t1;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QR_T(x) x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define QR_T(x)
#endif
#ifdef QR_WAVETIME
    unsigned long long wt_cells = 0;
#endif
    while (e != QR_NULL)
    {
        e = __builtin_amdgcn_readfirstlane(e);
#ifdef QR_WAVETIME
        wt_cells++;
#endif
        QR_T(t0);
        const CellS cs = ld_cell(E + e);
        const qr_elem el = cs.el;
#ifdef QR_STATS2
        asm volatile("" :: "s"(el.simd), "s"(el.next));
        QR_T(t1); tA += t1 - t0; nA++; t0 = t1;
#endif
#ifdef QR_STATS
        st_iter++; st_lanes += __popcll(__ballot(live && w.resume == QR_NULL));
#endif
        /*
         * Wave-level cull (ours, not in the reference): `bsph` holds a conservative world-space
         * bounding sphere of each surface's visible part (16 B per surface, scalar-cache
         * resident); if every live ray of the group provably misses it (perpendicular distance,
         * behind the origin, or beyond the current depth bound) the element cannot produce a hit
         * and is skipped without touching its 128-byte record.  Never applied to array /
         * bounding-volume cells or to a ray's own surface.
         */
        bool skip = false;
        int jump = QR_NULL;
        if ((el.kind & 4) && !QR_KNOB(SHADOW ? 64 : 128))
        {
            /* not reference arithmetic: fused operations are fine here.  The line misses the sphere iff
             * b^2 < dd * (|oc|^2 - R^2); 1e-5 * |oc|^2 * dd on the left absorbs the rounding of both
             * sides (a few 1e-7 relative to |oc|^2 * dd), on top of the inflated radius. */
            const float R = cs.r;
            const float ocx = cs.cx - r.org.x, ocy = cs.cy - r.org.y, ocz = cs.cz - r.org.z;
            const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
            const float oc2 = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
            const float R2 = R * R;
            const float q = oc2 - R2;
            const u32 outside = LM(q > 0.01f * R2);
            const u32 miss = (outside & (LM(__builtin_fmaf(oc2 * dd, 1e-5f, b * b) < dd * q) | LM(b < 0.0f)))
                           | LM(__builtin_fmaf(-R, dlen, b) > w.tbuf * dd);
            const u32 need = LM(live && w.resume == QR_NULL) & ~(miss & LM(el.simd != r.osi));
            skip = !__any(need != 0);
        }
#ifdef QR_STATS2
        QR_T(t1); tB += t1 - t0; t0 = t1;
#endif
        if (skip)
        {
#ifdef QR_STATS
            st_skip++;
#endif
            if (e == w.local_obj) w.local_obj = QR_NULL;
            if (w.resume == e) w.resume = QR_NULL;
        }
        else
        {
            jump = walk_element<SHADOW, false>(sc, e, el, D + el.simd, r, w, h, occluded, live
#ifdef QR_STATS2
                                 , seg
#endif
                                 );
#ifdef QR_STATS2
            QR_T(t1); tC += t1 - t0; nC++;
#endif
        }
        if (SHADOW && !__any(live)) break;
        e = jump != QR_NULL ? jump : el.next;
    }
#ifdef QR_WAVETIME
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        const size_t gw_ = (size_t)blockIdx.x * (QR_BLOCK / 64) + (threadIdx.x >> 6);
        unsigned long long *o = sc.stats + 28 + gw_ * QR_WT_SLOTS;
        o[SHADOW ? 4 : 5] += wt_cells; o[SHADOW ? 6 : 7] += 1;
    }
#endif
#ifdef QR_STATS2
    if (SHADOW && __ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        atomicAdd(&sc.stats[0], tA); atomicAdd(&sc.stats[1], tB); atomicAdd(&sc.stats[2], tC);
        atomicAdd(&sc.stats[3], nA); atomicAdd(&sc.stats[4], nC);
        atomicAdd(&sc.stats[5], seg[0]); atomicAdd(&sc.stats[6], seg[1]); atomicAdd(&sc.stats[7], seg[2]); atomicAdd(&sc.stats[8], seg[3]); atomicAdd(&sc.stats[9], seg[4]); atomicAdd(&sc.stats[10], seg[5]); atomicAdd(&sc.stats[11], seg[6]);
    }
#endif
#ifdef QR_STATS
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        const int b = SHADOW ? 0 : (r.osi == QR_NULL ? 3 : 6);
        atomicAdd(&sc.stats[b + 0], 1ull);
        atomicAdd(&sc.stats[b + 1], st_iter);
        atomicAdd(&sc.stats[b + 2], st_lanes);
        atomicAdd(&sc.stats[12 + b / 3], st_skip);
    }
#endif
}


/*
 * Divergent walk: every lane walks ITS OWN list at its own pace (element index, cell and surface
 * record are per-lane vector loads from L2 instead of wave-uniform scalar loads).  For incoherent
 * rays -- secondary rays of scenes with thousands of small objects, where a wave-packet walk visits
 * the union of what its rays need and keeps 5 of 64 lanes busy -- this does per-ray work only:
 * a ray that misses a bounding volume jumps straight behind the array, a ray whose bounding-sphere
 * test fails steps on alone.  Same per-ray semantics as walk_list (a packet of width one).
 */
template <bool SHADOW>
__device__ __forceinline__ void walk_div(const DevScene &sc, bool active, const Ray &r, Hit &h, bool &occluded)
{
    Walk w;
    w.dxyz = {0, 0, 0}; w.dijk = {0, 0, 0}; w.rijk = {0, 0, 0};
    w.tbuf = r.tmax;
    w.local_obj = QR_NULL;
    w.resume = QR_NULL;
    bool live = active;
    int e = active ? r.list : QR_NULL;
    const float dd = r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z;
    const float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
    while (__any(e != QR_NULL))
    {
        if (e != QR_NULL)
        {
            const DCell c = sc.elm[e];
            qr_elem el; el.simd = c.simd; el.data = c.data; el.next = c.next; el.kind = c.kind;
            bool skip = false;
            if (el.kind & 4)
            {
                const float ocx = c.cx - r.org.x, ocy = c.cy - r.org.y, ocz = c.cz - r.org.z;
                const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
                const float oc2 = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
                const float R2 = c.r * c.r;
                const float q = oc2 - R2;
                const bool outside = q > 0.01f * R2;
                const bool miss = (outside && (__builtin_fmaf(oc2 * dd, 1e-5f, b * b) < dd * q || b < 0.0f))
                               || __builtin_fmaf(-c.r, dlen, b) > w.tbuf * dd;
                skip = !(live && w.resume == QR_NULL) || (miss && el.simd != r.osi);
            }
            int jump = QR_NULL;
            if (skip)
            {
                if (e == w.local_obj) w.local_obj = QR_NULL;
                if (w.resume == e) w.resume = QR_NULL;
            }
            else
            {
#ifdef QR_STATS2
                unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
                jump = walk_element<SHADOW, true>(sc, e, el, sc.srf + el.simd, r, w, h, occluded, live
#ifdef QR_STATS2
                                                  , seg_
#endif
                                                  );
            }
            e = jump != QR_NULL ? jump : el.next;
            if (SHADOW && !live) e = QR_NULL;
        }
    }
}

/*
 * Wave-wide traversal: lanes with `active` walk their lists; lanes that share
 * a list head are walked together.
 */
template <bool SHADOW, bool DIV>
__device__ __forceinline__ void traverse(const DevScene &sc, bool active, const Ray &r, Hit &h, bool &occluded)
{
    h.t = r.tmax; h.si = QR_NULL; h.side = 0; h.loc = {0, 0, 0};
    occluded = false;
    active = active && r.list != QR_NULL;
    if (DIV)
    {
        walk_div<SHADOW>(sc, active, r, h, occluded);
        return;
    }
    unsigned long long pending = __ballot(active);
    while (pending != 0)
    {
        const int leader = __ffsll((long long)pending) - 1;
        const int head = __shfl(r.list, leader);
        const bool mine = active && r.list == head;
        pending &= ~__ballot(mine);
        if (mine)
        {
            walk_list<SHADOW>(sc, head, r, h, occluded);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* shading of the final hit, tracer.cpp:2166-3930 without the child packets  */
/* ------------------------------------------------------------------------ */

struct Frame
{
    float col[3];
    float c_trn, c_rfl, x0;
    float rdir[3];
    float hit[3];
    float loc[3];
    int   meta;             /* si << 4 | side << 3 | rf << 2 | phase (1 TR, 2 RF) */
};

struct Shaded
{
    V3 col;                 /* local colour after lights                      */
    V3 hit;                 /* world hit = child origin                       */
    V3 loc;                 /* local hit = child's ploc                       */
    V3 tdir;                /* refraction child direction (ctx_NEW after TR)  */
    V3 rdir;                /* reflection child direction (ctx_NEW after RF)  */
    float c_trn, c_rfl, x0;
    bool want_tr;           /* refraction child exists (M_TRN, not opaque)    */
    bool want_rf;           /* reflection pass applies (RF_ini reached)       */
    int  lst_tr, lst_rf;
};

struct Counters { u32 primary, shadow, reflect, refract; };

/* state of the enclosing recursion that only has to survive a shade() call */
struct Outer { V3 ret; int hit_id, sp, mode; };

template <bool COUNT, bool DIV>
__device__ __forceinline__ void shade(const DevScene &sc, bool act, const Ray &r, const Hit &h,
                                      Shaded &o, Counters &cnt)
{
    /* per-lane (divergent) material data; everything below is lane-private
     * except the wave-wide shadow traversals in the light loop */
    const int si = act ? h.si : 0;
    const int side = h.side;
    const DSurf *__restrict__ s = &sc.srf[si];
    const DShade *__restrict__ sd = &sc.shd[si];
    const FrmP fr = c_frm(sc);

    V3 nrm = {0, 0, 1};
    V3 tex = {0, 0, 0};
    V3 col = {0, 0, 0};
    V3 hit = {0, 0, 0};
    int props = 0;
    int mi = 0;
    int le = QR_NULL;

    if (act)
    {
        const float t = h.t;
        float x0, x1, x2, x3, x4, x5, x6;
        x4 = r.dir.x * t; hit.x = x4 + r.org.x;
        x5 = r.dir.y * t; hit.y = x5 + r.org.y;
        x6 = r.dir.z * t; hit.z = x6 + r.org.z;

        props = side | (side ? s->props1 : s->props0);
        mi = sd->mat[side];
        const u32 fl = s->flags;
        const u32 tside = side ? QR_SMASK : 0u;
        const int has_trm = (int)DF_TRM(fl);
        const int nkind = (int)DF_NKIND(fl);
        float tu = 0.0f, tv = 0.0f;
        V3 ln = {0, 0, 0};                          /* normal in surface space */

        if (nkind == 1)
        {
            /* PL_mat 4139-4193 */
            if (props & QR_PROP_TEXTURE)
            {
                tu = fxor(vget(h.loc, (int)DF_MAP(fl, 0)), DF_SGN(fl, 0));
                tv = fxor(vget(h.loc, (int)DF_MAP(fl, 1)), DF_SGN(fl, 1));
            }
            x6 = fxor(1.0f, tside);
            vset(ln, (int)DF_MAP(fl, 2), fxor(x6, DF_SGN(fl, 2)));
        }
        else
        {
            /* QD_mat 4845-4905 / TP_mat 4280-4336 */
            x4 = h.loc.x * s->sci[0]; x5 = h.loc.y * s->sci[1]; x6 = h.loc.z * s->sci[2];
            if (nkind == 2)
            {
                x4 = x4 - s->scj[0]; x5 = x5 - s->scj[1]; x6 = x6 - s->scj[2];
            }
            x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
            x1 = x1 + x2; x1 = x1 + x3;
            x0 = rsq(x1);
            x0 = fxor(x0, tside);
            ln.x = x4 * x0; ln.y = x5 * x0; ln.z = x6 * x0;
        }
        nrm = ln;
        if (has_trm != 0)
        {
            /* MT_nrm 2184-2263: transposed trnode matrix */
            const DSurf *__restrict__ tr = &sc.srf[s->trnode];
            const int ttrm = (int)DF_TRM(tr->flags);
            x1 = ln.x; x2 = ln.y; x3 = ln.z;
            x4 = tr->tci[0] * x1;
            x5 = tr->tcj[1] * x2;
            x6 = tr->tck[2] * x3;
            if (ttrm != 1)
            {
                x4 = x4 + tr->tcj[0] * x2;
                x4 = x4 + tr->tck[0] * x3;
                x5 = x5 + tr->tci[1] * x1;
                x5 = x5 + tr->tck[1] * x3;
                x6 = x6 + tr->tci[2] * x1;
                x6 = x6 + tr->tcj[2] * x2;
            }
            if (ttrm != 2)
            {
                x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
                x1 = x1 + x2; x1 = x1 + x3;
                x0 = rsq(x1);
                x4 = x4 * x0; x5 = x5 * x0; x6 = x6 * x0;
            }
            nrm.x = x4; nrm.y = x5; nrm.z = x6;
        }

        /* MT_tex 2293-2327, PAINT_FRAG / PAINT_COLX 653-673 */
        const qr_material *__restrict__ mt = &sc.mat[mi];
        u32 toff = 0;
        if (props & QR_PROP_TEXTURE)
        {
            x4 = mt->t_map[0] ? tv : tu;
            x5 = mt->t_map[1] ? tv : tu;
            x4 = x4 - mt->xoffs; x5 = x5 - mt->yoffs;
            x4 = x4 * mt->xscal; x5 = x5 * mt->yscal;
            const int32_t iu = cvt_floor(x4) & (int32_t)mt->xmask;
            const int32_t iv = cvt_floor(x5) & (int32_t)mt->ymask;
            toff = (u32)iu + ((u32)iv << (mt->yshft & 31));
        }
        const u32 texel = sc.texels[mt->tex + (int32_t)toff];
        const u32 cmask = mt->cmask;
        const float clampv = mt->clamp;
        tex.x = (float)(int32_t)((texel >> 16) & cmask) / clampv;
        tex.y = (float)(int32_t)((texel >> 8) & cmask) / clampv;
        tex.z = (float)(int32_t)(texel & cmask) / clampv;
        if (props & QR_PROP_GAMMA) { tex.x = tex.x * tex.x; tex.y = tex.y * tex.y; tex.z = tex.z * tex.z; }

        if (props & QR_PROP_LIGHT)
        {
            col = tex;                              /* LT_set */
        }
        else
        {
            col.x = tex.x * fr->amb[0];
            col.y = tex.y * fr->amb[1];
            col.z = tex.z * fr->amb[2];
            le = sd->lst[side * 2];
        }
    }

    /* lights, 2758-3156: wave-wide loop, per-lane light elements */
    while (__any(le != QR_NULL))
    {
        const bool has = le != QR_NULL;
        const DCell cel = sc.elm[has ? le : 0];
        qr_elem el; el.simd = cel.simd; el.data = cel.data; el.next = cel.next; el.kind = cel.kind;
        const qr_light *__restrict__ lg = &sc.lgt[has ? el.simd : 0];
        V3 L = {0, 0, 0};
        float dot = 0.0f;
        bool lm = false;
        if (has)
        {
            float x1, x2, x3, x0;
            x1 = lg->pos[0] - hit.x; L.x = x1; x1 = x1 * nrm.x;
            x2 = lg->pos[1] - hit.y; L.y = x2; x2 = x2 * nrm.y;
            x3 = lg->pos[2] - hit.z; L.z = x3; x3 = x3 * nrm.z;
            x0 = x1; x0 = x0 + x2; x0 = x0 + x3;
            dot = x0;
            lm = clt(0.0f, x0);
        }
        Ray sr;
        sr.org = hit; sr.dir = L; sr.tmin = 0.0f; sr.tmax = lg->t_max;
        sr.list = el.data; sr.osi = si; sr.oflg = side; sr.ploc = h.loc;
        Hit sh; bool occ;
        if (COUNT) { if (lm) cnt.shadow++; }
        if (QR_KNOB(2)) lm = false;
        if (QR_KNOB(1)) occ = false; else
        {
#ifdef QR_X_NOSHADOW
            occ = false; sh.si = 0;
#else
#ifdef QR_WAVETIME
            const unsigned long long wt_a = __builtin_amdgcn_s_memrealtime();
#endif
            traverse<true, DIV>(sc, lm, sr, sh, occ);
#ifdef QR_WAVETIME
            if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
                sc.stats[28 + ((size_t)blockIdx.x * (QR_BLOCK / 64) + (threadIdx.x >> 6)) * QR_WT_SLOTS + 10] += __builtin_amdgcn_s_memrealtime() - wt_a;
#endif
#endif
        }
        if (lm && !occ)
        {
            const qr_material *__restrict__ mt = &sc.mat[mi];
            float x0, x1, x2, x3, x4, x5, x6, x7;
            x1 = L.x; x4 = x1 * x1;
            x2 = L.y; x5 = x2 * x2;
            x3 = L.z; x6 = x3 * x3;
            x4 = x4 + x5; x4 = x4 + x6;
            const float r2 = x4;
            x0 = dot;
            if (props & QR_PROP_DIFFUSE)
            {
                x6 = x4;
                x5 = rsq(x4);
                x4 = x5 * x6;
                x6 = x6 * lg->a_qdr;
                x4 = x4 * lg->a_lnr;
                x6 = x6 + lg->a_cnt;
                x6 = x6 + x4;
                x4 = rsq(x6);
                x6 = x0;
                x0 = x0 * x4;
                x0 = x0 * x5;
                x0 = x0 * mt->l_dff;
            }
            else
            {
                x6 = x0;
                x0 = 0.0f;
            }
            bool plain = false;
            float spec = 0.0f;
            if (props & QR_PROP_SPECULAR)
            {
                x4 = x6; x5 = x6;
                x4 = x4 * nrm.x; x1 = x1 - x4; x1 = x1 - x4;
                x5 = x5 * nrm.y; x2 = x2 - x5; x2 = x2 - x5;
                x6 = x6 * nrm.z; x3 = x3 - x6; x3 = x3 - x6;
                x4 = r.dir.x; x1 = x1 * x4; x4 = x4 * x4;
                x5 = r.dir.y; x2 = x2 * x5; x5 = x5 * x5;
                x6 = r.dir.z; x3 = x3 * x6; x6 = x6 * x6;
                x6 = x6 + x4; x6 = x6 + x5;
                x1 = x1 + x2; x1 = x1 + x3;
                if (clt(0.0f, x1))
                {
                    x4 = r2;
                    x5 = rsq(x6); x1 = x1 * x5;
                    x5 = rsq(x4); x1 = x1 * x5;
                    /* fixed-point 28.4 power, 2981-3039 */
                    const u32 lpow = mt->l_pow;
                    u32 pw = lpow & 0xF;
                    x2 = x1; x4 = x1; x1 = 1.0f;
                    while (pw != 0)
                    {
                        x4 = __builtin_sqrtf(x4);
                        const u32 bit = pw & 0x8;
                        pw = (pw << 1) & 0xF;
                        if (bit) x1 = x1 * x4;
                    }
                    pw = lpow >> 4;
                    if (pw != 0)
                    {
                        x3 = x1; x1 = 1.0f;
                        do
                        {
                            const u32 bit = pw & 1;
                            pw >>= 1;
                            if (bit) x1 = x1 * x2;
                            x2 = x2 * x2;
                        }
                        while (pw != 0);
                        x1 = x1 * x3;
                    }
                    x1 = x1 * mt->l_spc;
                    if (props & QR_PROP_METAL) { x0 = x0 + x1; }
                    else { plain = true; spec = x1; }
                }
            }
            if (!plain)
            {
                x1 = tex.x * lg->col[0];
                x2 = tex.y * lg->col[1];
                x3 = tex.z * lg->col[2];
                x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
                col.x = x1 + col.x; col.y = x2 + col.y; col.z = x3 + col.z;
            }
            else
            {
                x7 = spec;
                x1 = tex.x; x2 = tex.y; x3 = tex.z;
                x4 = lg->col[0]; x5 = lg->col[1]; x6 = lg->col[2];
                x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
                x1 = x1 * x4; x2 = x2 * x5; x3 = x3 * x6;
                x4 = x4 * x7; x5 = x5 * x7; x6 = x6 * x7;
                x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
                col.x = x1 + col.x; col.y = x2 + col.y; col.z = x3 + col.z;
            }
        }
        le = has ? el.next : QR_NULL;
    }

    o.col = col; o.hit = hit; o.loc = h.loc;
    o.tdir = {0, 0, 0}; o.rdir = {0, 0, 0};
    o.c_trn = 0.0f; o.c_rfl = 0.0f; o.x0 = 0.0f;
    o.want_tr = false; o.want_rf = false;
    o.lst_tr = QR_NULL; o.lst_rf = QR_NULL;

    if (act)
    {
        const qr_material *__restrict__ mt = &sc.mat[mi];
        const float m_trn_c = mt->c_trn, m_rfl_c = mt->c_rfl;
        float c_trn = m_trn_c, c_rfl = m_rfl_c;
        float x0 = 0.0f, x1, x2, x3, x4 = 0.0f, x5, x6 = 0.0f, x7 = 0.0f;
        bool m_trn = true;

        /* transparency 3185-3552 */
        if (!(props & QR_PROP_OPAQUE))
        {
            const bool do_rfi = (props & QR_PROP_REFRACT) || (props & QR_PROP_FRESNEL);
            bool tir = false;
            V3 nd = r.dir;
            if (do_rfi)
            {
                x1 = r.dir.x; x7 = x1 * x1; x0 = x7;
                x2 = r.dir.y; x7 = x2 * x2; x0 = x0 + x7;
                x3 = r.dir.z; x7 = x3 * x3; x0 = x0 + x7;
                x7 = rsq(x0);
                x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
                x7 = x1 * nrm.x; x0 = x7;
                x7 = x2 * nrm.y; x0 = x0 + x7;
                x7 = x3 * nrm.z; x0 = x0 + x7;
                x4 = x0;
                x6 = mt->c_rfr;
                x0 = x0 * x6;
                x7 = x0 * x0;
                x7 = x7 + 1.0f;
                x7 = x7 - mt->rfr_2;
                if (props & QR_PROP_FRESNEL)
                {
                    m_trn = cle(0.0f, x7);
                    if (!m_trn)
                    {
                        c_trn = 0.0f;
                        c_rfl = m_rfl_c + m_trn_c;
                        tir = true;
                    }
                }
                if (!tir)
                {
                    x7 = __builtin_sqrtf(x7);
                    x0 = x0 + x7;
                    if (props & QR_PROP_REFRACT)
                    {
                        x5 = nrm.x * x0; x1 = x1 * x6; nd.x = x1 - x5;
                        x5 = nrm.y * x0; x2 = x2 * x6; nd.y = x2 - x5;
                        x5 = nrm.z * x0; x3 = x3 * x6; nd.z = x3 - x5;
                    }
                }
            }
            if (!tir)
            {
                if (props & QR_PROP_FRESNEL)
                {
                    x1 = x4;
                    x2 = x1; x2 = x2 * x6; x2 = x2 - x7;
                    x7 = x7 * x6;
                    x3 = x1;
                    x1 = x1 + x7;
                    x3 = x3 - x7;
                    x0 = x0 / x2;
                    x1 = x1 / x3;
                    x0 = x0 * x0; x1 = x1 * x1;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fabs_bits(x0);
                    const float f = x0 * m_trn_c;   /* m_trn is true here */
                    c_trn = m_trn_c - f;
                    c_rfl = m_rfl_c + f;
                }
                o.want_tr = m_trn;
                o.tdir = nd;
                o.lst_tr = sd->lst[(1 - side) * 2 + 1];
            }
        }

        /* TR_mix factor 3564-3573 */
        x0 = 1.0f - m_trn_c;
        x0 = x0 - m_rfl_c;
        x0 = cle(0.0f, x0) ? x0 : 0.0f;
        o.x0 = x0;

        /* reflections 3604-3815 */
        if ((props & QR_PROP_REFLECT) ||
            (!(props & QR_PROP_OPAQUE) && (props & QR_PROP_FRESNEL)))
        {
            x1 = r.dir.x; x4 = nrm.x; x7 = x1 * x1; x0 = x7;
            x2 = r.dir.y; x5 = nrm.y; x7 = x2 * x2; x0 = x0 + x7;
            x3 = r.dir.z; x6 = nrm.z; x7 = x3 * x3; x0 = x0 + x7;
            x7 = rsq(x0);
            x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
            x7 = x1 * x4; x0 = x7;
            x7 = x2 * x5; x0 = x0 + x7;
            x7 = x3 * x6; x0 = x0 + x7;
            x4 = x4 * x0; x1 = x1 - x4; x1 = x1 - x4; o.rdir.x = x1;
            x5 = x5 * x0; x2 = x2 - x5; x2 = x2 - x5; o.rdir.y = x2;
            x6 = x6 * x0; x3 = x3 - x6; x3 = x3 - x6; o.rdir.z = x3;

            if ((props & QR_PROP_FRESNEL) && (props & QR_PROP_OPAQUE))
            {
                if (props & QR_PROP_METAL)
                {
                    x6 = mt->c_rcp;
                    x4 = x0; x4 = x4 * x6; x4 = x4 + x4;
                    x0 = x0 * x0;
                    x6 = x6 * x6;
                    x6 = x6 + mt->ext_2;
                    x1 = x0; x1 = x1 * x6;
                    x0 = x0 + x6;
                    x1 = x1 + 1.0f;
                    x2 = x0; x3 = x1;
                    x0 = x0 + x4; x1 = x1 + x4;
                    x2 = x2 - x4; x3 = x3 - x4;
                    x0 = x0 / x2; x1 = x1 / x3;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fabs_bits(x0);
                }
                else
                {
                    x4 = x0;
                    x6 = mt->c_rfr;
                    x0 = x0 * x6;
                    x7 = x0 * x0;
                    x7 = x7 + 1.0f;
                    x7 = x7 - mt->rfr_2;
                    x7 = __builtin_sqrtf(x7);
                    x0 = x0 + x7;
                    x1 = x4;
                    x2 = x1; x2 = x2 * x6; x2 = x2 - x7;
                    x7 = x7 * x6;
                    x3 = x1;
                    x1 = x1 + x7;
                    x3 = x3 - x7;
                    x0 = x0 / x2; x1 = x1 / x3;
                    x0 = x0 * x0; x1 = x1 * x1;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fabs_bits(x0);
                }
                x0 = x0 - 1.0f;
                x0 = x0 * m_rfl_c;
                c_rfl = m_rfl_c + x0;
            }
            o.want_rf = true;
            o.lst_rf = sd->lst[side * 2 + 1];
        }
        o.c_trn = c_trn;
        o.c_rfl = c_rfl;
    }
}

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
