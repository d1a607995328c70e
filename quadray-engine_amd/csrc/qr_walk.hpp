/*
 * qr_walk.hpp - device code, list traversal over the COMPILED list programs of qr_program.h:
 *   clip()       CC_clp (tracer.cpp:1597-2160): depth + near test, hit point, conic-singularity fix, axis
 *                min/max, custom clipper program
 *   walk_list()  OO_cyc for the lanes of a wave that share one list: per cell a bounding-sphere cull (ours),
 *                then the cell's opcode: trnode transform (1419-1556), bounding volume (3955-4054), plane /
 *                quadric / two-plane solver (4062-4136, 4378-4842, 4216-4277), candidates through clip()
 *   traverse()   groups the lanes of a wave by list head
 * Everything about a cell and its surface is wave-uniform and arrives through scalar loads; what the
 * reference decides per ray from `ctx_LOCAL(OBJ)` while walking (which diff / ray a surface reads) is in the
 * cell's opcode (qr_compile.cpp).  A ray that misses a bounding volume records the offset at which it takes
 * part again (`resume`); one unsigned compare per cell gives the lanes that are on.
 * Included by qr_kernel.hpp after the shared types.
 */
#ifndef QR_WALK_HPP
#define QR_WALK_HPP

/*
 * Lane masks are kept as 64-bit scalars: a compare lands in an SGPR pair (`ballot` of a compare is the
 * v_cmp itself), logic is s_and / s_or / s_andn2, "any lane" is one s_cmp, and `lane_of` turns a mask back
 * into a per-lane predicate at no cost (the SGPR pair IS the predicate).  Masks only hold lanes that were
 * active where they were computed.
 */
typedef unsigned long long lm_t;
#if defined(QR_STATS) && defined(QR_GUARD)
/* diagnostic build: a cell offset that cannot be one -- not a multiple of the cell size, or beyond the image (DevHeader::img_bytes; a fixed
 * 256 MB bound until round 3 flagged the valid offsets of larger images) -- is recorded (stats[24..27]: tag, offset, previous offset,
 * count) and the walk ends */
#define QR_GUARD_POS(tag, p, prev, onbad) do { if (((p) & 31u) != 0u || (p) >= ((const QR_CONST DevHeader *)B)->img_bytes) { \
        if (atomicAdd(&stats[27], 1ull) == 0ull) { stats[24] = (tag); stats[25] = (p); stats[26] = (prev); } onbad; } } while (0)
#else
#define QR_GUARD_POS(tag, p, prev, onbad) do { } while (0)
#endif
#define LM(cond) __builtin_amdgcn_ballot_w64(cond)
#ifndef QR_CLIP_PREFETCH
#define QR_CLIP_PREFETCH 0  /* 1: clip() loads the next cell of a clipper program while the current one is evaluated.  Measured: no
                             * change for a band of the slowest footprints rendered alone (92.3 / 92.6 us), 1.6 % slower frames (two
                             * more spilled scalar registers): the cells of a program share cache lines, the waits are elsewhere */
#endif
__device__ __forceinline__ bool lane_of(lm_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ bool any_lane(bool b) { return LM(b) != 0ull; }

/*
 * The cell code below is shared by the wave-packet walk (cell wave-uniform: masks are scalar lane masks) and by
 * the per-lane walk (every lane its own cell: a "mask" is the lane's own predicate, because code under a
 * per-lane opcode test only sees part of the wave).  MK<DIV> is that choice.
 */
template <bool DIV> struct MK;
template <> struct MK<false>
{
    typedef lm_t T;
    static __device__ __forceinline__ T of(bool c) { return LM(c); }
    static __device__ __forceinline__ bool lane(T m) { return lane_of(m); }
    static __device__ __forceinline__ bool any(T m) { return m != 0ull; }
    static __device__ __forceinline__ T inv(T m) { return ~m; }
    static __device__ __forceinline__ T none() { return 0ull; }
};
template <> struct MK<true>
{
    typedef bool T;
    static __device__ __forceinline__ T of(bool c) { return c; }
    static __device__ __forceinline__ bool lane(T m) { return m; }
    static __device__ __forceinline__ bool any(T m) { return any_lane(m); }
    static __device__ __forceinline__ T inv(T m) { return !m; }
    static __device__ __forceinline__ T none() { return false; }
};

#ifdef QR_PROF
__device__ __forceinline__ unsigned long long qr_lanes(lm_t m) { return (unsigned long long)__popcll(m); }
__device__ __forceinline__ unsigned long long qr_lanes(bool m) { return (unsigned long long)__popcll(__ballot(m)); }
#endif

/* what clip() needs to know about the candidate's surface space */
template <bool DIV>
struct ClipIn
{
    V3 df, ry;                  /* the diff / ray the solver read (trnode space for QR_OPF_LOCAL cells)    */
    typename MK<DIV>::T dmask;  /* quadric: near-zero discriminant lanes (conic fix)                       */
    u32 amask;                  /* quadric: sign of `a`                                                    */
};

/* the first 80 bytes of a DSurf in SGPRs */
struct SurfS
{
    float pos0, pos1, pos2; u32 clip;
    float min0, min1, min2, d_eps;
    float max0, max1, max2, t_eps;
    float sci0, sci1, sci2, sci3;
    float scj0, scj1, scj2; u32 flags;
};

typedef u32 u32x16 __attribute__((ext_vector_type(16)));
/* one 64-byte + one 16-byte scalar load, issued back to back (a plane would do with 48 bytes, but a load that
 * depends on the opcode makes every field a loop-carried phi the compiler copies around) */
__device__ __forceinline__ void ld_surf(BaseP B, u32 off, SurfS &s)
{
    const u32x16 a = *(const QR_CONST u32x16 *)(B + off);
    const u32x4 c = *(const QR_CONST u32x4 *)(B + off + 64);
    s.pos0 = u2f(a.s0); s.pos1 = u2f(a.s1); s.pos2 = u2f(a.s2); s.clip = a.s3;
    s.min0 = u2f(a.s4); s.min1 = u2f(a.s5); s.min2 = u2f(a.s6); s.d_eps = u2f(a.s7);
    s.max0 = u2f(a.s8); s.max1 = u2f(a.s9); s.max2 = u2f(a.sa); s.t_eps = u2f(a.sb);
    s.sci0 = u2f(a.sc); s.sci1 = u2f(a.sd); s.sci2 = u2f(a.se); s.sci3 = u2f(a.sf);
    s.scj0 = u2f(c.x); s.scj1 = u2f(c.y); s.scj2 = u2f(c.z); s.flags = c.w;
}

/* 3x3 transform with the matrix rows of the DSurf at `off` (tracer.cpp:1447-1479 order) */
__device__ __forceinline__ V3 xform(BaseP B, u32 off, bool full, V3 in)
{
    const QR_CONST DSurf *p = (const QR_CONST DSurf *)(B + off);
    float x4 = p->tci[0] * in.x;
    float x5 = p->tcj[1] * in.y;
    float x6 = p->tck[2] * in.z;
    if (full)
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

__device__ __forceinline__ float sel3(float a, float b, float c, u32 i) { return i == 0 ? a : i == 1 ? b : c; }
/* component by one-hot axis flags (x, y, else z): two v_cndmask on scalar conditions */
__device__ __forceinline__ float axis3(const V3 &v, bool is_x, bool is_y) { return is_x ? v.x : (is_y ? v.y : v.z); }

/*
 * The eight compares of CC_clp that only narrow the mask (depth, near, axis min/max) as ONE chain of
 * v_cmpx: each compare ANDs itself into EXEC, so a condition costs one instruction instead of a compare
 * plus an s_and.  EXEC is saved and restored inside the block; predicates as in the reference
 * (cgt = NLE, cge = NLT: true on NaN; cle / clt: false on NaN).
 */
__device__ __forceinline__ lm_t clip_box(lm_t m, float tbuf, float tmin, float t, float x4, float x5, float x6, const SurfS &s)
{
    lm_t out, sv;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "v_cmpx_nle_f32 vcc, %3, %5\n\t"       /* cgt(tbuf, t)  */
                 "v_cmpx_lt_f32 vcc, %4, %5\n\t"        /* clt(tmin, t)  */
                 "v_cmpx_le_f32 vcc, %9, %6\n\t"        /* cle(min0, x4) */
                 "v_cmpx_nlt_f32 vcc, %12, %6\n\t"      /* cge(max0, x4) */
                 "v_cmpx_le_f32 vcc, %10, %7\n\t"
                 "v_cmpx_nlt_f32 vcc, %13, %7\n\t"
                 "v_cmpx_le_f32 vcc, %11, %8\n\t"
                 "v_cmpx_nlt_f32 vcc, %14, %8\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1"
                 : "=s"(out), "=&s"(sv)
                 : "s"(m), "v"(tbuf), "v"(tmin), "v"(t), "v"(x4), "v"(x5), "v"(x6),
                   "s"(s.min0), "s"(s.min1), "s"(s.min2), "s"(s.max0), "s"(s.max1), "s"(s.max2)
                 : "vcc");
    return out;
}

/* ------------------------------------------------------------------------ */
/* CC_clp, tracer.cpp:1597-2160.  Returns the lanes of `m` whose hit at `t`  */
/* survives; `loc` is the local hit (ctx_NEW_* of the surface's space).      */
/* ------------------------------------------------------------------------ */
template <bool DIV, bool CLIPL = false>
__device__ __forceinline__ typename MK<DIV>::T clip(BaseP B, const SurfS &s, u32 op, const Ray &r, float tbuf,
                                                    const ClipIn<DIV> &ci, float t, int side, typename MK<DIV>::T m, V3 &loc)
{
    typedef MK<DIV> K;
    QR_PROF_HIT(0);                     /* candidates through clip() */
    QR_FLOPS_M(15, qr_lanes(m));
    /* the opcode is re-read through an opaque copy: otherwise everything that only depends on it (the whole
     * axis decode of the conic fix, every flag as a 64-bit mask) is hoisted in front of the candidate loop and
     * paid by every cell */
    if constexpr (!DIV) asm volatile("" : "+s"(op));
    float x4, x5, x6;
    V3 hit;

    x4 = r.dir.x * t; x4 = x4 + r.org.x; hit.x = x4;
    x5 = r.dir.y * t; x5 = x5 + r.org.y; hit.y = x5;
    x6 = r.dir.z * t; x6 = x6 + r.org.z; hit.z = x6;

    if (op & QR_OPF_LOCAL)
    {
        /* the surface lives in a trnode's space */
        x4 = ci.ry.x * t; x4 = x4 + ci.df.x;
        x5 = ci.ry.y * t; x5 = x5 + ci.df.y;
        x6 = ci.ry.z * t; x6 = x6 + ci.df.z;
    }
    else
    {
        x4 = x4 - s.pos0;
        x5 = x5 - s.pos1;
        x6 = x6 - s.pos2;
    }
    V3 nw = {x4, x5, x6};                               /* the local hit: NEW[shift] */

    /* conic singularity solver, 1706-1856 */
    if (op & QR_OPF_CONIC)
    {
        u32 fl = s.flags;
        if constexpr (!DIV) asm volatile("" : "+s"(fl));
        const u32 conic = DF_CONIC(fl);
        const u32 mi = DF_MAP(fl, 0), mj = DF_MAP(fl, 1), mk = DF_MAP(fl, 2);
        float x0, x1, x2, x3;
        x1 = vget(nw, (int)mi); x1 = x1 * x1; x0 = x1;
        if (conic != 2) { x2 = vget(nw, (int)mj); x2 = x2 * x2; x0 = x0 + x2; }
        x3 = vget(nw, (int)mk); x3 = x3 * x3; x0 = x0 + x3;
        const typename K::T hm = K::of(clt(x0, s.t_eps)) & ci.dmask;
        if (K::any(hm))
        {
            QR_FLOPS_M(16, qr_lanes(hm));
            if (K::lane(hm))
            {
                const u32 sm = QR_SMASK;
                const float one = 1.0f;
                float r4;
                x2 = 0.0f;
                x1 = u2f((f2u(vget(ci.df, (int)mi)) & sm) ^ f2u(one));
                x3 = sel3(s.sci0, s.sci1, s.sci2, mi);
                r4 = one;
                if (conic != 2)
                {
                    x2 = u2f((f2u(vget(ci.df, (int)mj)) & sm) ^ f2u(one));
                    x3 = x3 + sel3(s.sci0, s.sci1, s.sci2, mj);
                    r4 = r4 + one;
                }
                x3 = x3 / sel3(s.sci0, s.sci1, s.sci2, mk);
                x3 = fxor(x3, sm);
                float y6 = x3;
                x3 = __builtin_sqrtf(x3);
                y6 = y6 + r4;
                r4 = rsq(y6);
                r4 = r4 * s.t_eps;
                x1 = x1 * r4; x2 = x2 * r4; x3 = x3 * r4;

                const u32 tside = side ? sm : 0u;
                x3 = fxor(x3, f2u(vget(ci.df, (int)mk)) & sm);
                x3 = fxor(x3, (tside & ci.amask) ^ ci.amask);
                const u32 u5 = (tside | ci.amask) ^ ci.amask;
                x1 = fxor(x1, u5);
                x2 = fxor(x2, u5);

                vset(nw, (int)mi, x1);
                if (conic != 2) vset(nw, (int)mj, x2);
                vset(nw, (int)mk, x3);
                x4 = nw.x; x5 = nw.y; x6 = nw.z;
            }
        }
    }
    loc = nw;

    /* depth + near test (1600-1640) and axis min/max (1874-1927): unclipped axes hold -inf / +inf, so the six
     * compares are unconditional (a lane still in `m` has a finite hit point) */
    if constexpr (DIV)
    {
        m = m && cgt(tbuf, t) && clt(r.tmin, t)
              && cle(s.min0, x4) && cge(s.max0, x4) && cle(s.min1, x5) && cge(s.max1, x5)
              && cle(s.min2, x6) && cge(s.max2, x6);
    }
    else m = clip_box(m, tbuf, r.tmin, t, x4, x5, x6, s);

    /* custom clipping, 1931-2151: the surface's clipper program.  Wave-uniform cells run it on scalar loads; the
     * per-lane walks never meet one (lists with clippers are not flagged for them, qr_compile.cpp) except the eager
     * path tracer's (CLIPL), where every lane runs the program of ITS cell */
    bool run_clip;
    if constexpr (DIV) run_clip = (op & QR_OPF_CLIP) && m; else run_clip = (op & QR_OPF_CLIP) && m != 0;
    if constexpr (!DIV || CLIPL) if (run_clip)
    {
        QR_PROF_HIT(1);                 /* clipper programs run */
        typename K::T c_acc = K::none();
        V3 cxyz = {0.0f, 0.0f, 0.0f};                   /* the hit in the cached clipper trnode's space */
        V3 pt = hit;                                    /* what a fast plane cell reads: the world hit, or cxyz while a trnode group lasts */
        u32 cp = s.clip;
#if QR_CLIP_PREFETCH
        /* the program's cells lie one behind the other: the next one is on its way while this one is evaluated (a lone
         * wave in the tail of a frame otherwise waits out one scalar-cache round trip per cell).  The cell behind the END
         * cell is read and dropped: programs are followed by other sections of the image or its tail padding. */
        if constexpr (!DIV) cp = __builtin_amdgcn_readfirstlane(cp);
        u32x8 cn = *(const QR_CONST u32x8 *)(B + cp);
#endif
        for (;;)
        {
            if constexpr (!DIV) cp = __builtin_amdgcn_readfirstlane(cp);
#if QR_CLIP_PREFETCH
            const u32x8 cc = cn;
            cp += (u32)sizeof(CClip);
            if (cc.s0 != 0) cn = *(const QR_CONST u32x8 *)(B + cp);
#else
            const u32x8 cc = *(const QR_CONST u32x8 *)(B + cp);
            cp += (u32)sizeof(CClip);
#endif
            const u32 cop = cc.s0;
            if (cop == 0) break;
            QR_PROF_HIT(2);             /* clipper cells */
            if (!(cop & (QR_CLT_ENTER | QR_CLT_LEAVE)))
                QR_FLOPS_M(((cop & QR_CLT_PLANE) ? 3 : ((cop & (QR_CLT_QUAD | QR_CLT_QUADJ)) ? 3 + 18 : 3))
                           + (((cop & (QR_CLT_TRNODE | QR_CLF_OWN)) && !(cop & QR_CLT_TRSAME)) ? ((cop & QR_CLF_FULLM) ? 15 : 3) : 0), qr_lanes(m));
            if (cop & QR_CLF_FASTPL)
            {
                QR_PROF_HIT(3);
                /* a plane that reads the hit as it stands: the component by the cell's axis masks, the sign folded in, one
                 * compare against the plane's position -- the decisions of +-(p_k - pos_k) <= 0 / >= 0 (qr_program.h) */
                u32 t = f2u(pt.x) & cc.s4;
                t = (f2u(pt.y) & cc.s5) | t;
                t = (f2u(pt.z) & cc.s6) | t;
                const float a = u2f(t ^ cc.s3), val = u2f(cc.s2);
                const typename K::T keep_le = K::of(cle(a, val)), keep_ge = K::of(cge(a, val));
                m = m & ((cop & QR_CLF_INNER) ? keep_ge : keep_le);
            }
            else if (cop & (QR_CLT_ENTER | QR_CLT_LEAVE))
            {
                if (cop & QR_CLT_ENTER) { c_acc = m; m = (cop & QR_CLF_CDEF) ? K::inv(K::none()) : K::none(); }
                else m = K::inv(m) & c_acc;
            }
            else
            {
                const u32 koff = cc.s1;
                const u32x4 k0 = *(const QR_CONST u32x4 *)(B + koff);
                const float kp0 = u2f(k0.x), kp1 = u2f(k0.y), kp2 = u2f(k0.z);
                if (cop & (QR_CLT_TRSAME | QR_CLT_TRNODE))
                {
                    QR_PROF_HIT(4);
                    if (cop & QR_CLT_TRSAME)
                    {
                        /* the clipper trnode is the surface's own: its local hit + pos is the hit in that space */
                        cxyz.x = x4 + s.pos0; cxyz.y = x5 + s.pos1; cxyz.z = x6 + s.pos2;
                    }
                    else
                    {
                        V3 d;
                        d.x = hit.x - kp0; d.y = hit.y - kp1; d.z = hit.z - kp2;
                        cxyz = xform(B, koff, (cop & QR_CLF_FULLM) != 0, d);
                    }
                    pt = cxyz;
                }
                else
                {
                    V3 cv;
                    if (cop & QR_CLF_CACHED)
                    {
                        cv.x = cxyz.x - kp0; cv.y = cxyz.y - kp1; cv.z = cxyz.z - kp2;
                    }
                    else
                    {
                        V3 d;
                        d.x = hit.x - kp0; d.y = hit.y - kp1; d.z = hit.z - kp2;
                        cv = (cop & QR_CLF_OWN) ? xform(B, koff, (cop & QR_CLF_FULLM) != 0, d) : d;
                    }
                    float f4;
                    QR_PROF_HIT((cop & QR_CLT_PLANE) ? 5 : 6);
                    if (cop & QR_CLT_PLANE)
                    {
                        f4 = fxor(axis3(cv, (cop & QR_CLF_KX) != 0, (cop & QR_CLF_KY) != 0), (cop & QR_CLF_SGNK) ? QR_SMASK : 0u);
                    }
                    else
                    {
                        const u32x4 k1 = *(const QR_CONST u32x4 *)(B + koff + 48);
                        const float ks0 = u2f(k1.x), ks1 = u2f(k1.y), ks2 = u2f(k1.z), ks3 = u2f(k1.w);
                        float f5, f6;
                        if (cop & QR_CLT_QUADJ)
                        {
                            const u32x4 k2 = *(const QR_CONST u32x4 *)(B + koff + 64);
                            float f1, f2, f3;
                            f4 = cv.x; f1 = u2f(k2.x); f1 = f1 + f1; f1 = f1 * f4;
                            f4 = f4 * f4; f4 = f4 * ks0; f4 = f4 - f1;
                            f5 = cv.y; f2 = u2f(k2.y); f2 = f2 + f2; f2 = f2 * f5;
                            f5 = f5 * f5; f5 = f5 * ks1; f5 = f5 - f2;
                            f6 = cv.z; f3 = u2f(k2.z); f3 = f3 + f3; f3 = f3 * f6;
                            f6 = f6 * f6; f6 = f6 * ks2; f6 = f6 - f3;
                        }
                        else
                        {
                            f4 = cv.x; f4 = f4 * f4; f4 = f4 * ks0;
                            f5 = cv.y; f5 = f5 * f5; f5 = f5 * ks1;
                            f6 = cv.z; f6 = f6 * f6; f6 = f6 * ks2;
                        }
                        f4 = f4 - ks3; f4 = f4 + f5; f4 = f4 + f6;
                    }
                    m = m & ((cop & QR_CLF_INNER) ? K::of(cge(f4, 0.0f)) : K::of(cle(f4, 0.0f)));
                }
            }
            if (cop & QR_CLF_LASTC) pt = hit;           /* the trnode group ends here */
        }
    }
    return m;
}

/* AR_ptr 3955-4054: does the line hit the array's bounding volume? */
__device__ __forceinline__ bool bv_hit(const V3 &ry, const V3 &df, float sci0, float sci1, float sci2, float sci3)
{
    float x0, x1, x2, x3, x4, x5, x6, x7;
    x1 = ry.x; x0 = sci0 * x1; x5 = df.x; x7 = sci0 * x5;
    x3 = x1; x1 = x1 * x0; x3 = x3 * x7; x5 = x5 * x7;
    x2 = ry.y; x0 = sci1 * x2; x6 = df.y; x7 = sci1 * x6;
    x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
    x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
    x2 = ry.z; x0 = sci2 * x2; x6 = df.z; x7 = sci2 * x6;
    x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
    x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
    x5 = x5 - sci3;
    x5 = x5 * x1;
    x3 = x3 * x3;
    x3 = x3 - x5;
    return cle(0.0f, x3);
}

#ifdef QR_WAVETIME
__shared__ unsigned qr_wt_groups[2];    /* list groups walked by this wave: nearest-hit, shadow (tools/gpu_wavetime.py) */
__shared__ unsigned long long qr_wt_shadow;     /* 100 MHz ticks spent in shadow traversals */
__shared__ unsigned qr_wt_cells[4];     /* packet walks of this wave: cells looked at (nearest-hit, shadow), cells solved (nearest-hit, shadow) */
#endif

/* per-lane state of one list walk */
struct WalkState
{
    V3 txyz, trijk;         /* trnode cache: diff and ray in the trnode's space                             */
    float tbuf, tbd;        /* ctx_T_BUF and tbuf * |dir|^2 (cull)                                          */
    u32 resume;             /* packet walk: the lane takes part in cells at offsets >= resume;
                             * 0xFFFFFFFF: shadow ray occluded, the ray has left the walk                   */
};

/*
 * One cell with a solver for the lanes that are active: diff / ray in the surface's space, the solver,
 * candidate roots through clip(), depth write (or occlusion for shadow rays).
 * DIV = false: `op`, `srf_off` and the record `s` are wave-uniform (SGPRs);
 * DIV = true : every lane brings its own cell and record (the per-lane walk); cells with clipper programs are
 *              not allowed here.
 */
/* diff / ray in a cell's space, tracer.cpp:1352-1556: world, cached trnode space, or the surface's own transform */
__device__ __forceinline__ void cell_space(BaseP B, u32 op, u32 srf_off, float pos0, float pos1, float pos2,
                                           const Ray &r, const WalkState &w, V3 &df, V3 &ry)
{
    if (op & QR_OPF_CACHED)
    {
        df.x = w.txyz.x - pos0; df.y = w.txyz.y - pos1; df.z = w.txyz.z - pos2;
        ry = w.trijk;
    }
    else
    {
        df.x = r.org.x - pos0; df.y = r.org.y - pos1; df.z = r.org.z - pos2;
        ry = r.dir;
        if (op & QR_OPF_OWN)
        {
            df = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, df);
            ry = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, r.dir);
        }
    }
}

template <bool SHADOW, bool DIV, bool WORLD = false, bool CLIPL = false>
__device__ __forceinline__ void solve_cell(BaseP B, u32 op, u32 srf_off, const SurfS &s, const Ray &r,
                                           float dd, WalkState &w, Hit &h)
{
    typedef MK<DIV> K;
    typedef typename K::T mask_t;
    /* ---- diff / ray in the surface's space, 1352-1556 (WORLD: the list has world-space cells only) ---- */
    ClipIn<DIV> ci;
    ci.dmask = K::none(); ci.amask = 0;
    if constexpr (WORLD)
    {
        ci.df.x = r.org.x - s.pos0; ci.df.y = r.org.y - s.pos1; ci.df.z = r.org.z - s.pos2;
        ci.ry = r.dir;
    }
    else cell_space(B, op, srf_off, s.pos0, s.pos1, s.pos2, r, w, ci.df, ci.ry);
    /* a secondary ray on its own surface starts from the parent's local hit, 1352-1373 */
    const mask_t same = K::of(srf_off == r.osrf);
    if (K::any(same))
    {
        const bool sl = K::lane(same);
        ci.df.x = sl ? r.ploc.x : ci.df.x; ci.df.y = sl ? r.ploc.y : ci.df.y; ci.df.z = sl ? r.ploc.z : ci.df.z;
    }
    const V3 ry = ci.ry, df = ci.df;

    {
        /* up to two candidate roots per lane, in the lane's own order */
        float ct0 = 0.0f, ct1 = 0.0f;
        int   cs0 = 0, cs1 = 0;
        mask_t cm0 = K::none(), cm1 = K::none();
        int   ncand = 0;

        /* diff 3, own transform of diff and ray 2 x (3 or 15), solver 1 / 31 + 10 / 21 */
        QR_FLOPS(3 + ((op & QR_OPF_OWN) ? ((op & QR_OPF_FULLM) ? 30 : 6) : 0) + ((op & QR_OPT_PLANE) ? 1 : ((op & QR_OPT_QUADRIC) ? 41 : 21)));
        QR_PROF_HIT((op & QR_OPT_PLANE) ? 8 : ((op & QR_OPT_QUADRIC) ? 9 : 10));
        QR_PROF_HIT(SHADOW ? 11 : 12);
        if (op & (QR_OPF_OWN | QR_OPF_CACHED)) QR_PROF_HIT(13);
        if (op & QR_OPF_CONIC) QR_PROF_HIT(14);
        if (op & QR_OPT_PLANE)
        {
            /* PL_ptr 4062-4136 */
            const bool kx = (op & QR_OPF_KX) != 0, ky = (op & QR_OPF_KY) != 0;
            const u32 sg = (op & QR_OPF_SGNK) ? QR_SMASK : 0u;
            const float dk = fxor(axis3(df, kx, ky), sg ^ QR_SMASK);
            const float rk = fxor(axis3(ry, kx, ky), sg);
            /* Pre-test (ours): the hit only survives clip() if t_min < t < t_buf.  With t_min >= 0 a
             * quotient of opposite signs cannot, and |dk| >= |rk| * t_buf * (1 + 2^-20) means
             * t >= t_buf whatever the rounding of the division; dropping those lanes here changes
             * nothing, and when no lane is left the wave skips the IEEE division and clip(). */
            const mask_t opposite = K::of(((f2u(dk) ^ f2u(rk)) & QR_SMASK) != 0);
            const mask_t beyond = K::of(fabs_bits(dk) >= fabs_bits(rk) * (w.tbuf * 1.000001f));
            cm0 = K::of(cne(0.0f, rk)) & K::inv(same) & K::inv((opposite | beyond) & K::of(r.tmin >= 0.0f));
            if (K::any(cm0)) { ct0 = dk / rk; ncand = 1; }
            cs0 = clt(rk, 0.0f) ? 0 : 1;
        }
        else
        {
            float a, b, cq, d;
            if (op & QR_OPT_QUADRIC)
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
                a = x1; b = x4; cq = x6; d = x3;
            }
            else
            {
                /* TP_ptr 4216-4277 */
                const bool ix = (op & QR_OPF_IX) != 0, iy = (op & QR_OPF_IY) != 0;
                const bool kx = (op & QR_OPF_KX) != 0, ky = (op & QR_OPF_KY) != 0;
                const V3 sc3 = {s.sci0, s.sci1, s.sci2};
                float x0, x1, x2, x3, x4, x5, x6, x7;
                x1 = axis3(ry, ix, iy); x5 = axis3(df, ix, iy); x3 = axis3(sc3, ix, iy);
                x2 = axis3(ry, kx, ky); x6 = axis3(df, kx, ky); x4 = axis3(sc3, kx, ky);
                x0 = x5; x7 = x6;
                x6 = x6 * x1; x5 = x5 * x2; x5 = x5 - x6; x5 = x5 * x5; x5 = x5 * x3; x5 = x5 * x4;
                x5 = fabs_bits(x5);
                x6 = x3; x3 = x3 * x0; x4 = x4 * x7; x3 = x3 * x1; x4 = x4 * x2; x3 = x3 + x4;
                x4 = axis3(sc3, kx, ky);
                x0 = x0 * x0; x7 = x7 * x7; x0 = x0 * x6; x7 = x7 * x4; x0 = x0 + x7;
                x1 = x1 * x1; x2 = x2 * x2; x1 = x1 * x6; x2 = x2 * x4; x1 = x1 + x2;
                a = x1; b = x3; cq = x0; d = x5;
            }

            /* QD_rts 4449-4658 */
            const u32 sm = QR_SMASK;
            const mask_t xmask = K::of(cle(0.0f, d));
            /* CHECK_MASK(OO_end, NONE, xmask), 4455 */
            if (K::any(xmask))
            {
                b = fxor(b, sm);
                const mask_t dmask = xmask & K::of(clt(d, s.d_eps));
                ci.dmask = dmask;

                const float sd = fxor(__builtin_sqrtf(d), sm & f2u(b));
                const float bd = b + sd;
                const bool m_pos = cle(0.0f, sd);
                /* m_neg = cgt(0, sd) = !m_pos unless sd is NaN (then both selections come out 0) */
                const bool m_neg = cgt(0.0f, sd);
                const float t2n = u2f((m_neg ? f2u(cq) : 0u) | (m_pos ? f2u(bd) : 0u));
                const float t1n = u2f((m_neg ? f2u(bd) : 0u) | (m_pos ? f2u(cq) : 0u));
                float t2d = u2f((m_neg ? f2u(bd) : 0u) | (m_pos ? f2u(a) : 0u));
                float t1d = u2f((m_neg ? f2u(a) : 0u)  | (m_pos ? f2u(bd) : 0u));
                a = u2f((m_pos ? f2u(a) : 0u) | (m_neg ? f2u(a) : 0u));

                const u32 amask = sm & f2u(a);
                ci.amask = amask;
                if (K::any(dmask))
                {
                    if (K::lane(dmask))
                    {
                        if (ceq(t1n, 0.0f)) t1d = 1.0f;
                        if (ceq(t2n, 0.0f)) t2d = 1.0f;
                    }
                }
                float t1 = t1n / t1d;
                float t2 = t2n / t2d;
                const mask_t t1msk = K::of(cne(t1d, 0.0f));
                const mask_t t2msk = K::of(cne(t2d, 0.0f));
                if (K::any(dmask))
                {
                    if (K::lane(dmask))
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
                        tdf = K::lane(t1msk & t2msk) ? tdf : 0.0f;
                        t1 = t1 + tdf;
                        t2 = t2 - tdf;
                    }
                }

                const bool inner_first = cgt(0.0f, a);      /* only read where xmask holds */
                mask_t mo = xmask & t1msk, mi2 = xmask & t2msk;
                if (K::any(same))
                {
                    /* CHECK_SIDE 531-540: on its own surface a ray that left through the outer side
                     * (flags 0: reflected off it, 3: passed through from inside) skips the inner root,
                     * one that left through the inner side (1, 2) the outer root */
                    const mask_t so = K::of(((r.oflg ^ (r.oflg >> 1)) & 1) != 0);
                    mo = mo & K::inv(same & so);
                    mi2 = mi2 & K::inv(same & K::inv(so));
                }
                ncand = 2;
                ct0 = inner_first ? t2 : t1; ct1 = inner_first ? t1 : t2;
                cs0 = inner_first ? 1 : 0;   cs1 = inner_first ? 0 : 1;
                const mask_t inf = K::of(inner_first);
                cm0 = (inf & mi2) | (K::inv(inf) & mo); cm1 = (inf & mo) | (K::inv(inf) & mi2);
            }
        }

        mask_t done = K::none();
        /* per-lane cells: every lane brings its own number of candidates, so both slots are visited (a slot that
         * was not filled has an empty mask) */
        const int np = DIV ? 2 : ncand;
#pragma nounroll
        for (int p = 0; p < np; p++)
        {
            const float t = p == 0 ? ct0 : ct1;
            const int side = p == 0 ? cs0 : cs1;
            mask_t m = (p == 0 ? cm0 : cm1) & K::inv(done);
            if (!K::any(m)) continue;
            V3 loc;
            m = clip<DIV, CLIPL>(B, s, op, r, w.tbuf, ci, t, side, m, loc);
            done = done | m;
            if (K::lane(m))
            {
                if (SHADOW)
                {
                    /* CHECK_SHAD 549-589: by the surface's material (static class in the opcode) */
                    u32 opl = op;
                    if constexpr (!DIV) asm volatile("" : "+s"(opl));
                    bool casts = (opl & QR_OPF_NOSHAD) == 0;
                    if (opl & QR_OPF_SIDESHAD)
                    {
                        const QR_CONST DSurf *P = (const QR_CONST DSurf *)(B + srf_off);
                        const int props = side ? P->props1 : P->props0;
                        casts = !((props & QR_PROP_LIGHT) || ((props & QR_PROP_TRANSP) && !(props & QR_PROP_REFRACT)));
                    }
                    if (casts) w.resume = 0xFFFFFFFFu;        /* occluded: the ray leaves the walk */
                }
                else
                {
                    /* PAINT_FRAG 653-662: depth write; shading is deferred */
                    w.tbuf = t; w.tbd = t * dd;
                    h.t = t; h.srf = srf_off; h.side = side;
                    h.loc = loc;
                }
            }
        }
        if (!K::any(done)) { QR_PROF_HIT(26); if (op & QR_OPT_PLANE) QR_PROF_HIT(27); if (ncand == 0) QR_PROF_HIT(28); }      /* nobody hit anything */
    }
        }

/*
 * OO_cyc for the lanes of a wave that share the list program at `head` (wave-uniform, not 0).
 * SHADOW: any-hit walk, ends as soon as every ray is occluded.
 */
#ifndef QR_ASM_CULL
#define QR_ASM_CULL 1       /* 0: the run of culled cells as compiled C++ (the statistics / profile builds always use that form; the guard build keeps the assembly: QR_GUARD_ASM) */
#endif

/*
 * The run of culled cells of walk_list, written out: load a cell, leave unless it is a solver cell with a cull bound, test the
 * bound (box slabs or sphere, the arithmetic of the C++ form below), leave if some ray that is on needs the cell, step.
 * hipcc 7.2 compiles the C++ loop to 25 scalar instructions a cell (structured-control-flow bookkeeping: s_mov -1 / s_andn2 vcc,
 * exec / s_cbranch_vccnz chains) around 19 vector ones; a third of a frame's instructions are scalar and two thirds of those
 * sat in this loop.  Here a culled cell costs 12-14 scalar instructions.  The cell lives in s[88:95] (an inline-asm operand
 * has no sub-register syntax, and the fields are needed one by one), scratch masks in s[96:99]; the cell is handed out in
 * four 64-bit operands when the run ends.
 * Box test: miss <=> tn > tf | tf < 0 | tn > tb1 with tb1 = 1.00001 * depth bound >= 0, folded into ONE compare:
 * max(tn, 0) > min(tf, tb1)  (max / min drop a NaN operand exactly where the three compares were false).
 */
template <bool BOXC>
__device__ __forceinline__ void cull_run(BaseP B, u32 &pos, u32x8 &c, const Ray &r, const WalkState &w,
                                         float idx, float idy, float idz, float nox, float noy, float noz, float dd, float dde, float dlen)
{
    unsigned long long c01, c23, c45, c67;
    float t0, t1, t2, t3, t4, t5, t6;
    const float tb1 = w.tbuf * 1.00001f;
    u32 cur = __builtin_amdgcn_readfirstlane(pos);           /* wave-uniform by construction; says so to the compiler */
    if constexpr (BOXC)
    {
        asm volatile(
            "qr_cull_top_%=:\n"
            "s_load_dwordx8 s[88:95], %[B], %[pos] offset:0x0\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_and_b32 s96, s88, 40\n"                      /* QR_OPF_CULL | QR_OPT_BV */
            "s_cmp_lg_u32 s96, 32\n"
            "s_cbranch_scc1 qr_cull_out_%=\n"
            "s_bitcmp0_b32 s88, 19\n"                       /* QR_OPF_BOX */
            "s_cbranch_scc1 qr_cull_sph_%=\n"
            "v_fma_f32 %[t0], s90, %[idx], %[nox]\n"
            "v_fma_f32 %[t1], s93, %[idx], %[nox]\n"
            "v_fma_f32 %[t2], s91, %[idy], %[noy]\n"
            "v_fma_f32 %[t3], s94, %[idy], %[noy]\n"
            "v_fma_f32 %[t4], s92, %[idz], %[noz]\n"
            "v_fma_f32 %[t5], s95, %[idz], %[noz]\n"
            "v_min_f32 %[t6], %[t0], %[t1]\n"
            "v_max_f32 %[t0], %[t0], %[t1]\n"
            "v_min_f32 %[t1], %[t2], %[t3]\n"
            "v_max_f32 %[t2], %[t2], %[t3]\n"
            "v_min_f32 %[t3], %[t4], %[t5]\n"
            "v_max_f32 %[t4], %[t4], %[t5]\n"
            "v_max3_f32 %[t6], %[t6], %[t1], %[t3]\n"
            "v_min3_f32 %[t0], %[t0], %[t2], %[t4]\n"
            "v_max_f32 %[t6], 0, %[t6]\n"
            "v_min_f32 %[t0], %[t0], %[tb1]\n"
            "v_cmp_gt_f32 vcc, %[t6], %[t0]\n"
            "qr_cull_tst_%=:\n"
            "v_cmp_ne_u32 s[96:97], s89, %[osrf]\n"
            "s_and_b64 s[96:97], vcc, s[96:97]\n"
            "v_cmp_ge_u32 vcc, %[pos], %[res]\n"
            "s_andn2_b64 s[96:97], vcc, s[96:97]\n"         /* rays that are on and not provably missed (or whose own surface it is) */
            "s_cbranch_scc1 qr_cull_out_%=\n"
            "s_add_u32 %[pos], %[pos], 32\n"
            "s_branch qr_cull_top_%=\n"
            "qr_cull_sph_%=:\n"
            "v_sub_f32 %[t0], s92, %[ox]\n"
            "v_sub_f32 %[t1], s93, %[oy]\n"
            "v_sub_f32 %[t2], s94, %[oz]\n"
            "v_mul_f32 %[t3], %[t0], %[dx]\n"
            "v_mul_f32 %[t4], %[t0], %[t0]\n"
            "v_fmac_f32 %[t3], %[t1], %[dy]\n"
            "v_fmac_f32 %[t4], %[t1], %[t1]\n"
            "v_fmac_f32 %[t3], %[t2], %[dz]\n"
            "v_fmac_f32 %[t4], %[t2], %[t2]\n"
            "v_subrev_f32 %[t0], s90, %[t4]\n"
            "v_mul_f32_e64 %[t1], %[t3], |%[t3]|\n"
            "v_cmp_lt_f32 vcc, s91, %[t4]\n"
            "v_fmac_f32 %[t1], %[t4], %[dde]\n"
            "v_mul_f32 %[t0], %[dd], %[t0]\n"
            "v_fma_f32 %[t2], -s95, %[dlen], %[t3]\n"
            "v_cmp_lt_f32 s[96:97], %[t1], %[t0]\n"
            "v_cmp_gt_f32 s[98:99], %[t2], %[tbd]\n"
            "s_and_b64 vcc, vcc, s[96:97]\n"
            "s_or_b64 vcc, vcc, s[98:99]\n"
            "s_branch qr_cull_tst_%=\n"
            "qr_cull_out_%=:\n"
            "s_mov_b64 %[c01], s[88:89]\n"
            "s_mov_b64 %[c23], s[90:91]\n"
            "s_mov_b64 %[c45], s[92:93]\n"
            "s_mov_b64 %[c67], s[94:95]\n"
            : [c01] "=&s"(c01), [c23] "=&s"(c23), [c45] "=&s"(c45), [c67] "=&s"(c67), [pos] "+s"(cur),
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6)
            : [B] "s"(B), [idx] "v"(idx), [idy] "v"(idy), [idz] "v"(idz), [nox] "v"(nox), [noy] "v"(noy), [noz] "v"(noz), [tb1] "v"(tb1),
              [ox] "v"(r.org.x), [oy] "v"(r.org.y), [oz] "v"(r.org.z), [dx] "v"(r.dir.x), [dy] "v"(r.dir.y), [dz] "v"(r.dir.z),
              [dd] "v"(dd), [dde] "v"(dde), [dlen] "v"(dlen), [tbd] "v"(w.tbd), [res] "v"(w.resume), [osrf] "v"(r.osrf)
            : "vcc", "scc", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    }
    else
    {
        (void)idx; (void)idy; (void)idz; (void)nox; (void)noy; (void)noz; (void)tb1; (void)t5; (void)t6;
        asm volatile(
            "qr_cull_top_%=:\n"
            "s_load_dwordx8 s[88:95], %[B], %[pos] offset:0x0\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_and_b32 s96, s88, 0x80028\n"                 /* QR_OPF_CULL | QR_OPT_BV | QR_OPF_BOX: a box cell is not culled here */
            "s_cmp_lg_u32 s96, 32\n"
            "s_cbranch_scc1 qr_cull_out_%=\n"
            "v_sub_f32 %[t0], s92, %[ox]\n"
            "v_sub_f32 %[t1], s93, %[oy]\n"
            "v_sub_f32 %[t2], s94, %[oz]\n"
            "v_mul_f32 %[t3], %[t0], %[dx]\n"
            "v_mul_f32 %[t4], %[t0], %[t0]\n"
            "v_fmac_f32 %[t3], %[t1], %[dy]\n"
            "v_fmac_f32 %[t4], %[t1], %[t1]\n"
            "v_fmac_f32 %[t3], %[t2], %[dz]\n"
            "v_fmac_f32 %[t4], %[t2], %[t2]\n"
            "v_subrev_f32 %[t0], s90, %[t4]\n"
            "v_mul_f32_e64 %[t1], %[t3], |%[t3]|\n"
            "v_cmp_lt_f32 vcc, s91, %[t4]\n"
            "v_fmac_f32 %[t1], %[t4], %[dde]\n"
            "v_mul_f32 %[t0], %[dd], %[t0]\n"
            "v_fma_f32 %[t2], -s95, %[dlen], %[t3]\n"
            "v_cmp_lt_f32 s[96:97], %[t1], %[t0]\n"
            "v_cmp_gt_f32 s[98:99], %[t2], %[tbd]\n"
            "s_and_b64 vcc, vcc, s[96:97]\n"
            "s_or_b64 vcc, vcc, s[98:99]\n"
            "v_cmp_ne_u32 s[96:97], s89, %[osrf]\n"
            "s_and_b64 s[96:97], vcc, s[96:97]\n"
            "v_cmp_ge_u32 vcc, %[pos], %[res]\n"
            "s_andn2_b64 s[96:97], vcc, s[96:97]\n"
            "s_cbranch_scc1 qr_cull_out_%=\n"
            "s_add_u32 %[pos], %[pos], 32\n"
            "s_branch qr_cull_top_%=\n"
            "qr_cull_out_%=:\n"
            "s_mov_b64 %[c01], s[88:89]\n"
            "s_mov_b64 %[c23], s[90:91]\n"
            "s_mov_b64 %[c45], s[92:93]\n"
            "s_mov_b64 %[c67], s[94:95]\n"
            : [c01] "=&s"(c01), [c23] "=&s"(c23), [c45] "=&s"(c45), [c67] "=&s"(c67), [pos] "+s"(cur),
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)
            : [B] "s"(B),
              [ox] "v"(r.org.x), [oy] "v"(r.org.y), [oz] "v"(r.org.z), [dx] "v"(r.dir.x), [dy] "v"(r.dir.y), [dz] "v"(r.dir.z),
              [dd] "v"(dd), [dde] "v"(dde), [dlen] "v"(dlen), [tbd] "v"(w.tbd), [res] "v"(w.resume), [osrf] "v"(r.osrf)
            : "vcc", "scc", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    }
    pos = cur;
    c.s0 = (u32)c01; c.s1 = (u32)(c01 >> 32); c.s2 = (u32)c23; c.s3 = (u32)(c23 >> 32);
    c.s4 = (u32)c45; c.s5 = (u32)(c45 >> 32); c.s6 = (u32)c67; c.s7 = (u32)(c67 >> 32);
}

/* BOXC: the walk knows box cull cells (QR_OPF_BOX).  The kernel instance with the per-lane walks is compiled without: images it
 * serves carry none (qr_compile.cpp), and when it is forced onto one (QR_DIV=1) such a cell is simply not culled */
template <bool SHADOW, bool BOXC>
__device__ __forceinline__ void walk_list(BaseP B, u32 head, const Ray &r, Hit &h, bool &occluded
#ifdef QR_STATS
                                          , unsigned long long *stats
#endif
                                          )
{
    WalkState w;
    w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0};
    w.tbuf = r.tmax;
    w.resume = 0;
    const float dd = r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z;
    /* only the cull uses the ray length: an upper bound is enough there, so the 1-instruction
     * approximate square root (1 ulp) inflated by 2^-20 replaces the IEEE expansion */
    const float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
    const float dde = dd * 1e-5f;
    w.tbd = w.tbuf * dd;
    /* box cull cells (QR_OPF_BOX): slab test with the reciprocal direction; a zero component becomes +-1e-30 (the ray does not
     * move along that axis: both slab distances come out huge with the sign of the side the origin is on) */
    float idx = 0.0f, idy = 0.0f, idz = 0.0f, nox = 0.0f, noy = 0.0f, noz = 0.0f;
    if (BOXC && (c_frm(B)->img_flags & QR_IMG_BOXES))
    {
        const float dxs = __builtin_fabsf(r.dir.x) < 1e-30f ? __builtin_copysignf(1e-30f, r.dir.x) : r.dir.x;
        const float dys = __builtin_fabsf(r.dir.y) < 1e-30f ? __builtin_copysignf(1e-30f, r.dir.y) : r.dir.y;
        const float dzs = __builtin_fabsf(r.dir.z) < 1e-30f ? __builtin_copysignf(1e-30f, r.dir.z) : r.dir.z;
        idx = __builtin_amdgcn_rcpf(dxs); idy = __builtin_amdgcn_rcpf(dys); idz = __builtin_amdgcn_rcpf(dzs);
        /* slab distance (bound - org) / dir as one fused operation, bound * id - org * id: its error, an ulp of |org * id|, is
         * 6e-8 |org| in space -- the boxes are padded by 2e-6 of the scene's largest coordinate on top of their own margin */
        nox = -(r.org.x * idx); noy = -(r.org.y * idy); noz = -(r.org.z * idz);
    }
    u32 pos = __builtin_amdgcn_readfirstlane(head);
    QR_PROF_HIT(SHADOW ? 18 : 19);      /* packet walks */
    QR_CULL_FLOPS((BOXC && (c_frm(B)->img_flags & QR_IMG_BOXES)) ? 15 : 9);
#ifdef QR_STATS
    unsigned long long st_iter = 0, st_lanes = 0, st_skip = 0;
#endif
    for (;;)
    {
        /*
         * Wave-level cull (ours, not in the reference): a solver cell carries a conservative world-space bounding
         * sphere of the surface's visible part; if every ray that is on provably misses it (perpendicular distance /
         * behind the origin, or beyond the current depth bound) the element cannot produce a hit and is skipped
         * without touching its record.  Never applied to a ray's own surface.  Not reference arithmetic: fused
         * operations are fine here.  With the origin outside the sphere (|oc|^2 > 1.01 R^2) the line misses it iff
         * b < 0 or b^2 < dd (|oc|^2 - R^2); both in one compare with b |b|; 1e-5 |oc|^2 dd on the left absorbs the
         * rounding of both sides (a few 1e-7 relative to |oc|^2 dd), on top of the inflated radius.
         * Runs of culled cells stay in this small loop (half of all cells of the demo scenes end here); it holds
         * the walk's only cell load.
         */
        u32x8 c;
#if QR_ASM_CULL && ((!defined(QR_STATS) && !defined(QR_PROF) && !defined(QR_WAVETIME) && !defined(QR_GUARD)) || defined(QR_GUARD_ASM))
        /* QR_GUARD_ASM (the guard library of the GPU suite): the hand-written loop stays -- the guarded walk runs the product's
         * instructions -- and the cursor is checked before the run loads its first cell and when it hands a cell out (the
         * statistics of culled cells are not kept in that build) */
        QR_GUARD_POS(1, pos, head, return);
        cull_run<BOXC>(B, pos, c, r, w, idx, idy, idz, nox, noy, noz, dd, dde, dlen);
        QR_GUARD_POS(4, pos, head, return);
#else
        for (;;)
        {
            /* `pos` is wave-uniform by construction; every assignment says so to the compiler (readfirstlane of the
             * new value), which keeps the cursor in an SGPR: with one readfirstlane here instead, it lived in a VGPR
             * and made the round trip v_mov / v_readfirstlane once per cell */
            QR_GUARD_POS(1, pos, head, return);
#ifdef QR_WAVETIME
            if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) qr_wt_cells[SHADOW ? 1 : 0]++;
#endif
            c = *(const QR_CONST u32x8 *)(B + pos);
            QR_PROF_HIT(16);            /* cells loaded by packet walks */
            if ((c.s0 & (QR_OPF_CULL | QR_OPT_BV | (BOXC ? 0u : QR_OPF_BOX))) != QR_OPF_CULL) break;
            lm_t miss;
            if (BOXC && (c.s0 & QR_OPF_BOX))
            {
                /* axis-aligned box {lo, hi} of the surface's visible part (inflated at upload, far above the rounding here):
                 * the ray misses it when it leaves one slab before it has entered all three, when the box lies behind the
                 * origin, or when it lies beyond the depth bound */
                const float ax = __builtin_fmaf(u2f(c.s2), idx, nox), bx = __builtin_fmaf(u2f(c.s5), idx, nox);
                const float ay = __builtin_fmaf(u2f(c.s3), idy, noy), by = __builtin_fmaf(u2f(c.s6), idy, noy);
                const float az = __builtin_fmaf(u2f(c.s4), idz, noz), bz = __builtin_fmaf(u2f(c.s7), idz, noz);
                const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
                const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
                miss = LM(tn > tf) | LM(tf < 0.0f) | LM(tn > w.tbuf * 1.00001f);
                QR_PROF_HIT(29);
                QR_CULL_FLOPS(22);
            }
            else
            {
                QR_CULL_FLOPS(20);
                const float R = u2f(c.s7), R2 = u2f(c.s2), R2x = u2f(c.s3);
                const float ocx = u2f(c.s4) - r.org.x, ocy = u2f(c.s5) - r.org.y, ocz = u2f(c.s6) - r.org.z;
                const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
                const float oc2 = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
                const float q = oc2 - R2;
                miss = (LM(oc2 > R2x) & LM(__builtin_fmaf(oc2, dde, b * __builtin_fabsf(b)) < dd * q))
                     | LM(__builtin_fmaf(-R, dlen, b) > w.tbd);
            }
            const lm_t need = LM(w.resume <= pos) & ~(miss & LM(c.s1 != r.osrf));
#ifdef QR_STATS
            st_iter++; st_lanes += __popcll(LM(w.resume <= pos));
#endif
            if (need != 0) break;
            QR_PROF_HIT(17);            /* culled without looking at the surface */
#ifdef QR_STATS
            st_skip++;
#endif
            pos = __builtin_amdgcn_readfirstlane(pos + 32);
        }
#endif
        const u32 op = c.s0;
        if (op == 0) break;
        const u32 srf_off = c.s1;
        const lm_t on = LM(w.resume <= pos);
        u32 next = pos + 32;
        bool full = true;
#ifdef QR_STATS
        if (!((op & (QR_OPF_CULL | QR_OPT_BV)) == QR_OPF_CULL)) { st_iter++; st_lanes += __popcll(on); }
#endif
        lm_t far = 0;                           /* bounding volume: lanes for which the whole array is out of reach */
        if ((op & (QR_OPF_CULL | QR_OPT_BV)) == (QR_OPF_CULL | QR_OPT_BV))
        {
            /* the array's own conservative sphere (qr_compile.cpp): entirely behind the origin, or entirely beyond
             * the current depth bound -> nothing inside can be hit by this ray: treated as a miss of the volume */
            const float R = u2f(c.s7);
            const float ocx = u2f(c.s4) - r.org.x, ocy = u2f(c.s5) - r.org.y, ocz = u2f(c.s6) - r.org.z;
            const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
            far = on & (LM(__builtin_fmaf(R, dlen, b) < 0.0f) | LM(__builtin_fmaf(-R, dlen, b) > w.tbd));
            if (far != 0) { if (lane_of(far)) w.resume = c.s2; }
            full = (on & ~far) != 0;
        }

        if (full)
        {
            if (op & QR_OPT_TRNODE)
            {
                QR_PROF_HIT(20);
                QR_FLOPS_M(3 + ((op & QR_OPF_FULLM) ? 30 : 6), __popcll(on));
                /* array element with a transform: diff and ray in its space, cached for the surfaces behind it */
                if (lane_of(on))
                {
                    const u32x4 p0 = *(const QR_CONST u32x4 *)(B + srf_off);
                    V3 d;
                    d.x = r.org.x - u2f(p0.x); d.y = r.org.y - u2f(p0.y); d.z = r.org.z - u2f(p0.z);
                    w.txyz = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, d);
                    w.trijk = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, r.dir);
                }
            }
            else if (op & QR_OPT_BV)
            {
                /* AR_ptr 3955-4054; the volume travels with the cell (CBvExt) */
                QR_PROF_HIT(21);
                QR_FLOPS_M(3 + 25, __popcll(on & ~far));
                next = pos + 64;
                const u32x8 x = *(const QR_CONST u32x8 *)(B + pos + 32);
                if (lane_of(on & ~far))
                {
                    V3 df, ry;
                    cell_space(B, op, srf_off, u2f(x.s0), u2f(x.s1), u2f(x.s2), r, w, df, ry);
                    if (!bv_hit(ry, df, u2f(x.s4), u2f(x.s5), u2f(x.s6), u2f(x.s7))) w.resume = c.s2;   /* back at the array's end */
                }
                /* the reference jumps a whole packet behind an array whose bounding volume no lane hits
                 * (tracer.cpp:4040-4054); rays that were off already wait for the end of an enclosing array,
                 * which lies at or behind this array's end (arrays nest, qr_compile.cpp) */
                if (LM(w.resume <= pos) == 0) next = c.s2;
            }
            else
            {
                SurfS s;
                ld_surf(B, srf_off, s);
#ifdef QR_WAVETIME
                if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) qr_wt_cells[SHADOW ? 3 : 2]++;
#endif
                if (lane_of(on)) solve_cell<SHADOW, false>(B, op, srf_off, s, r, dd, w, h);
                if (SHADOW)
                {
                    if (LM(w.resume != 0xFFFFFFFFu) == 0) break;      /* every ray of the group is occluded */
                }
            }
        }
        if (!full) next = c.s2;                 /* every ray that was on is out of the array's reach: nobody enters it */
        pos = __builtin_amdgcn_readfirstlane(next);
    }
    if (SHADOW) occluded = w.resume == 0xFFFFFFFFu;
#ifdef QR_STATS
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        const int b = SHADOW ? 0 : (r.osrf == 0 ? 3 : 6);
        atomicAdd(&stats[b + 0], 1ull);
        atomicAdd(&stats[b + 1], st_iter);
        atomicAdd(&stats[b + 2], st_lanes);
        atomicAdd(&stats[12 + b / 3], st_skip);
    }
#endif
}

/* SurfS of a per-lane cell: five 16-byte vector loads */
__device__ __forceinline__ void ld_surf_lane(BaseP B, u32 off, SurfS &s)
{
    const u32x4 a = *(const QR_CONST u32x4 *)(B + off), b = *(const QR_CONST u32x4 *)(B + off + 16),
                c = *(const QR_CONST u32x4 *)(B + off + 32), d = *(const QR_CONST u32x4 *)(B + off + 48),
                e = *(const QR_CONST u32x4 *)(B + off + 64);
    s.pos0 = u2f(a.x); s.pos1 = u2f(a.y); s.pos2 = u2f(a.z); s.clip = a.w;
    s.min0 = u2f(b.x); s.min1 = u2f(b.y); s.min2 = u2f(b.z); s.d_eps = u2f(b.w);
    s.max0 = u2f(c.x); s.max1 = u2f(c.y); s.max2 = u2f(c.z); s.t_eps = u2f(c.w);
    s.sci0 = u2f(d.x); s.sci1 = u2f(d.y); s.sci2 = u2f(d.z); s.sci3 = u2f(d.w);
    s.scj0 = u2f(e.x); s.scj1 = u2f(e.y); s.scj2 = u2f(e.z); s.flags = e.w;
}

/* per-lane form of the cell cull of walk_list: c0 = {op, srf, R^2, 1.01 R^2}, c1 = {cx, cy, cz, R} */
__device__ __forceinline__ bool div_culled(const u32x4 &c0, const u32x4 &c1, const Ray &r, float dd, float dde, float dlen, float tbd)
{
    QR_CULL_FLOPS(20);
    const float R = u2f(c1.w), R2 = u2f(c0.z), R2x = u2f(c0.w);
    const float ocx = u2f(c1.x) - r.org.x, ocy = u2f(c1.y) - r.org.y, ocz = u2f(c1.z) - r.org.z;
    const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
    const float oc2 = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
    const float q = oc2 - R2;
    const bool miss = ((oc2 > R2x) & (__builtin_fmaf(oc2, dde, b * __builtin_fabsf(b)) < dd * q))
                    | (__builtin_fmaf(-R, dlen, b) > tbd);
    return miss & (c0.y != r.osrf);
}

#ifndef QR_DIV_BATCH
#define QR_DIV_BATCH 16     /* per-lane walk: solve as soon as this many lanes hold a candidate cell (24: +3 %, 32: +6 % frame time) */
#endif

/*
 * PER-LANE walk for incoherent rays: every lane walks ITS OWN list at its own pace (cells and records through
 * vector loads).  A wave-packet walk visits the union of what its rays need -- for the secondary rays of a scene
 * with thousands of small objects that is 1300 cells with 5 of 64 lanes interested in each -- while here a ray
 * that misses a bounding volume jumps behind the array alone and a ray whose bounding-sphere test fails steps on
 * alone.  The walk alternates two phases so that both run with most lanes busy: STEP (lanes without a candidate
 * advance one cell: END, trnode transform, bounding volume, cull test) and SOLVE (lanes that stand on a cell
 * they may hit run solver + clip, all together, once QR_DIV_BATCH of them wait or nobody can step any more).
 * Per ray the cells are met in list order, so depth-test ties resolve as in the packet walk; results are the same
 * (a packet of width one).  Only for lists without clipper programs (QR_LISTF_DIV, set by the compiler).
 */
template <bool SHADOW>
__device__ __forceinline__ void walk_div(BaseP B, bool active, const Ray &r, Hit &h, bool &occluded
#ifdef QR_STATS
                                         , unsigned long long *stats
#endif
                                         )
{
    WalkState w;
    w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0};
    w.tbuf = r.tmax;
    w.resume = 0;
    const float dd = r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z;
    const float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
    const float dde = dd * 1e-5f;
    w.tbd = w.tbuf * dd;
    u32 pos = active ? (r.list & ~31u) : 0u;    /* 0: the lane has finished */
    u32 p_op = 0, p_srf = 0;                    /* the candidate cell the lane stands on (p_op == 0: none) */
#ifdef QR_STATS
    unsigned long long st_iter = 0, st_lanes = 0, st_solve = 0, st_slanes = 0;
#endif
    for (;;)
    {
        const lm_t pend = LM(p_op != 0);
        const lm_t adv = LM(pos != 0) & ~pend;
        if ((adv | pend) == 0) break;
        if (adv != 0 && __popcll(pend) < QR_DIV_BATCH)
        {
            /* ---- STEP ----
             * one 64-byte load per lane: the cell and what follows it -- the extension of a bounding-volume cell, or
             * the next cell, which is then handled in the same round trip when the first one was culled (the walk is
             * bound by the latency of these dependent loads, not by arithmetic) */
#ifdef QR_STATS
            st_iter++; st_lanes += __popcll(adv);
#endif
            if (lane_of(adv))
            {
                QR_GUARD_POS(2, pos, r.list, return);
                const u32x4 a0 = *(const QR_CONST u32x4 *)(B + pos), a1 = *(const QR_CONST u32x4 *)(B + pos + 16),
                            b0 = *(const QR_CONST u32x4 *)(B + pos + 32), b1 = *(const QR_CONST u32x4 *)(B + pos + 48);
                const u32 op = a0.x, srf_off = a0.y;
                u32 next = pos + 32;
                if (op == 0) next = 0;
                else if (op & QR_OPT_SOLVER)
                {
                    /* (a box cull cell, QR_OPF_BOX, holds no sphere: the per-lane walks do not cull on it) */
                    if ((op & (QR_OPF_CULL | QR_OPF_BOX)) != QR_OPF_CULL || !div_culled(a0, a1, r, dd, dde, dlen, w.tbd)) { p_op = op; p_srf = srf_off; }
                    else if ((b0.x & QR_OPT_SOLVER) != 0)
                    {
                        /* second cell of the load */
                        next = pos + 64;
                        if ((b0.x & (QR_OPF_CULL | QR_OPF_BOX)) != QR_OPF_CULL || !div_culled(b0, b1, r, dd, dde, dlen, w.tbd)) { p_op = b0.x; p_srf = b0.y; }
                    }
                }
                else if (op & QR_OPT_BV)
                {
                    next = pos + 64;
                    bool far = false;
                    if (op & QR_OPF_CULL)
                    {
                        /* the array cull of walk_list: entirely behind the origin or beyond the depth bound */
                        const float R = u2f(a1.w);
                        const float ocx = u2f(a1.x) - r.org.x, ocy = u2f(a1.y) - r.org.y, ocz = u2f(a1.z) - r.org.z;
                        const float bb = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
                        far = (__builtin_fmaf(R, dlen, bb) < 0.0f) | (__builtin_fmaf(-R, dlen, bb) > w.tbd);
                    }
                    if (far) next = a0.z;
                    else
                    {
                        QR_FLOPS(3 + 25);
                        V3 df, ry;
                        cell_space(B, op, srf_off, u2f(b0.x), u2f(b0.y), u2f(b0.z), r, w, df, ry);
                        if (!bv_hit(ry, df, u2f(b1.x), u2f(b1.y), u2f(b1.z), u2f(b1.w))) next = a0.z;
                    }
                }
                else
                {
                    /* trnode: diff and ray in its space, cached for the surfaces behind it */
                    const u32x4 p0 = *(const QR_CONST u32x4 *)(B + srf_off);
                    V3 d;
                    d.x = r.org.x - u2f(p0.x); d.y = r.org.y - u2f(p0.y); d.z = r.org.z - u2f(p0.z);
                    w.txyz = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, d);
                    w.trijk = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, r.dir);
                }
                pos = next;
            }
        }
        else
        {
            /* ---- SOLVE ---- */
#ifdef QR_STATS
            st_solve++; st_slanes += __popcll(pend);
#endif
            if (lane_of(pend))
            {
                SurfS s;
                ld_surf_lane(B, p_srf, s);
                solve_cell<SHADOW, true>(B, p_op, p_srf, s, r, dd, w, h);
                p_op = 0;
                if (SHADOW) { if (w.resume == 0xFFFFFFFFu) pos = 0; }
            }
        }
    }
    if (SHADOW) { if (w.resume == 0xFFFFFFFFu) occluded = true; }   /* only the lanes that walked here */
#ifdef QR_STATS
    const unsigned long long st_start = (unsigned long long)__popcll(__ballot(active));
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        if (SHADOW) { atomicAdd(&stats[9], 1ull); atomicAdd(&stats[10], st_iter); atomicAdd(&stats[11], st_lanes);
                      atomicAdd(&stats[15], st_start); atomicAdd(&stats[22], st_solve); }
        else        { atomicAdd(&stats[16], 1ull); atomicAdd(&stats[17], st_iter); atomicAdd(&stats[18], st_lanes);
                      atomicAdd(&stats[19], st_solve); atomicAdd(&stats[20], st_slanes);
                      atomicAdd(&stats[21], st_start); }
    }
#endif
}


/* ------------------------------------------------------------------------------------------------------------ */
/* walk_pool: the per-lane walk with WORK HAND-OVER between the lanes of the wave.                              */
/* ------------------------------------------------------------------------------------------------------------ */
#ifndef QR_LONG_MIN
#define QR_LONG_MIN 1       /* rays on one long hierarchy from which on they walk per lane: even one (its walk is shared out by walk_pool; 4: +5 % frame time) */
#endif
#ifndef QR_INCOH_MIN
#define QR_INCOH_MIN 3      /* lanes left when the leader's group is at most a third of them: walk per lane (measured 3 vs 12: -7 % frame time) */
#endif
#ifndef QR_INCOH_RATIO
#define QR_INCOH_RATIO 3    /* per lane when the leader's group is at most QR_INCOH_DEN / QR_INCOH_RATIO of the lanes left */
#endif
#ifndef QR_INCOH_DEN
#define QR_INCOH_DEN 1
#endif
#ifndef QR_POOL
#define QR_POOL 1          /* 0: walk_div without hand-over (A/B) */
#endif
#ifndef QR_POOL_MIN_IDLE
#define QR_POOL_MIN_IDLE 12     /* hand-over round as soon as this many lanes have nothing to do (2: +2 %, 6: +1.5 % frame time) */
#endif
#ifndef QR_POOL_FLAT
#define QR_POOL_FLAT 1             /* flat lists are cut in halves for idle lanes (0: hand-over only at the boundaries bounding volumes give) */
#endif
#ifndef QR_POOL_FLAT_MIN_BYTES
#define QR_POOL_FLAT_MIN_BYTES 128u    /* a flat range of at least four cells gives its far half away */
#endif
#ifndef QR_POOL_FLAT_MIN_IDLE
#define QR_POOL_FLAT_MIN_IDLE 8
#endif
#ifndef QR_POOL_MIN_BYTES
#define QR_POOL_MIN_BYTES 384u  /* a lane only gives a range of at least this many bytes of cells away */
#endif

/* LDS of the one-wave workgroup used by walk_pool (one object for both instantiations) */
struct PoolLds
{
    unsigned long long key[64];     /* per ray (= owner lane).  Nearest hit: (t bits << 32) | offset of the hit's cell, the
                                     * minimum over everything walked for that ray so far; shadow: != 0 once occluded      */
    u32x4 hit[64];                  /* payload of the hit `key` stands for: {srf | side, loc}                              */
    u32x4 give[64];                 /* hand-over slots: {owner, first cell, end of range, -}                               */
};
__device__ __forceinline__ PoolLds &pool_lds() { __shared__ PoolLds p; return p; }

__device__ __forceinline__ int lanes_below(lm_t m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

/*
 * walk_div leaves most of the wave idle on a long hierarchy: rays end at very different times (a shadow ray is
 * occluded after 5 cells or walks 400), measured 6.9 (shadow) / 15.7 (secondary) of 64 lanes stepping on the 10 000
 * object scene.  Here a lane's work is a RANGE of its ray's list program [pos, iend), and a lane that has entered a
 * bounding volume knows a cell boundary ahead of it (`give`: the start of a later child of that volume, stored by
 * the compiler in the volume's cell, CBvExt::mid) from which on the rest of its range can be walked by somebody
 * else: every enclosing volume has been entered, the list has no trnode state (the compiler only stores boundaries
 * where that holds), so the cells from there on mean the same to any lane that holds the ray.  As soon as
 * QR_POOL_MIN_IDLE lanes are idle, lanes with such a boundary hand [give, iend) to them (slots in LDS, the ray itself
 * through ds_bpermute from its owner's registers) and keep [pos, give).
 * Results meet in LDS per ray: a shadow ray's occlusion flag (everybody on that ray stops), or the minimum of
 * (t, cell offset) over the hits of all ranges of that ray -- the reference takes the first cell in list order among
 * equal depths (strict compare, tracer.cpp:1626), which is exactly that minimum, so the order in which ranges are
 * walked does not matter.  Depth tests inside a range use the range's own bound; the best depth known for the ray when
 * the range was taken only feeds the conservative sphere culls.
 */
/*
 * One conservative test for either kind of cell (ours, not the reference's; a cell without QR_OPF_CULL carries R = +inf
 * and fails none of it).  c1 = {centre, R}; r2 = the sphere's radius^2 the discriminant is taken against (solver cell:
 * R^2; sphere volume: the volume's own sci_w), out2 = "origin outside" threshold (solver: 1.01 R^2; sphere volume: -1,
 * always; other volume: +inf, never).  Solver cells use the RAY (b |b|: a sphere behind the origin is missed),
 * volumes the LINE like AR_ptr (b b), plus `behind`.  Also returns the discriminant terms: a sphere volume is
 * certainly entered when b2 - m >= rhs.
 */
__device__ __forceinline__ bool pool_cull(u32 srf, const u32x4 &c1, float r2, float out2, bool line, const Ray &r,
                                          float dd, float dde, float dlen, float tbd, float &b2, float &m, float &rhs)
{
    QR_CULL_FLOPS(22);
    const float R = u2f(c1.w);
    const float ocx = u2f(c1.x) - r.org.x, ocy = u2f(c1.y) - r.org.y, ocz = u2f(c1.z) - r.org.z;
    const float b = __builtin_fmaf(ocz, r.dir.z, __builtin_fmaf(ocy, r.dir.y, ocx * r.dir.x));
    const float oc2 = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
    b2 = (line ? __builtin_fabsf(b) : b) * __builtin_fabsf(b);
    m = oc2 * dde;
    rhs = dd * (oc2 - r2);
    const bool miss = ((oc2 > out2) & (b2 + m < rhs))
                    | (__builtin_fmaf(-R, dlen, b) > tbd) | (__builtin_fmaf(R, dlen, b) < 0.0f);
    return miss & (srf != r.osrf);
}

template <bool SHADOW>
__device__ __forceinline__ void walk_pool(BaseP B, bool active, const Ray &r, Hit &h, bool &occluded
#ifdef QR_STATS
                                          , unsigned long long *stats
#endif
                                          )
{
    PoolLds &P = pool_lds();
    const int lane = (int)(threadIdx.x & 63u);
    Ray cr = r;                                 /* the ray this lane walks for (its own to begin with) */
    int owner = lane;
    WalkState w;
    w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0};
    w.tbuf = cr.tmax;
    w.resume = 0;
    float dd = cr.dir.x * cr.dir.x + cr.dir.y * cr.dir.y + cr.dir.z * cr.dir.z;
    float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
    float dde = dd * 1e-5f;
    float cull_t = cr.tmax;                     /* best depth known for the ray when this range was taken */
    w.tbd = w.tbuf * dd;
    u32 pos = active ? (r.list & ~31u) : 0u;    /* 0: nothing to walk */
    u32 iend = 0xFFFFFFFFu;                     /* end of the range; the first range of a ray ends at the END cell */
    u32 give = 0;
#if QR_POOL_FLAT
    /* a FLAT list (no bounding-volume cell: every 32 bytes a cell) tells its length in the word in front of it (qr_compile.cpp):
     * its range can be cut anywhere, and is -- in halves, whenever lanes idle (round 4: the shadow lists of hits on small
     * objects are ten such cells, and a walk lasted as long as its slowest ray's ten) */
    bool flat = false;
    if (pos != 0)
    {
        const u32 lw = *(const QR_CONST u32 *)(B + (pos - 4u));
        flat = (lw & 1u) != 0;
        if (flat) iend = pos + (lw & ~31u);
    }
#endif
    u32 p_op = 0, p_srf = 0, p_pos = 0;         /* the candidate cell the lane stands on (p_op == 0: none) */
    bool busy = active;
#if defined(QR_STATS) && defined(QR_GUARD)
    u32 g_prev = 0, g_how = 0;
#define QR_G(x) x
#else
#define QR_G(x)
#endif
    Hit lh; lh.t = 0.0f; lh.srf = 0; lh.side = 0; lh.loc = {0, 0, 0};
    u32 lh_pos = 0;
    P.key[lane] = SHADOW ? 0ull : (((unsigned long long)f2u(r.tmax) << 32) | 0xFFFFFFFFull);
    __syncthreads();
#ifdef QR_STATS
    unsigned long long st_iter = 0, st_lanes = 0, st_solve = 0, st_slanes = 0, st_give = 0;
#endif
    for (;;)
    {
        const lm_t pend = LM(p_op != 0);
        const lm_t adv = LM(pos != 0) & ~pend;
        const lm_t work = adv | pend;
        /* ranges that ended: their best hit meets the ray's */
        const lm_t fin = LM(busy) & ~work;
        if (fin != 0)
        {
            const bool f = lane_of(fin);
            if (f) busy = false;
            if (!SHADOW)
            {
                const bool hv = f && lh.srf != 0;
                const unsigned long long k = ((unsigned long long)f2u(lh.t) << 32) | (unsigned long long)lh_pos;
                if (hv) atomicMin(&P.key[owner], k);
                __syncthreads();
                if (hv && P.key[owner] == k)
                    P.hit[owner] = u32x4{lh.srf | (u32)lh.side, f2u(lh.loc.x), f2u(lh.loc.y), f2u(lh.loc.z)};
            }
        }
        if (work == 0) break;

        /* ---- hand-over ---- */
#if QR_POOL_FLAT
        /* flat range: the boundary is its middle cell */
        if (flat && pos != 0 && p_op == 0) give = (iend - pos) >= QR_POOL_FLAT_MIN_BYTES ? pos + (((iend - pos) >> 6) << 5) : 0u;
        const bool can_give = pos != 0 && give > pos && (iend - give) >= (flat ? 32u : QR_POOL_MIN_BYTES);
        const lm_t givers = LM(can_give);
        const lm_t idle = ~work;
        if (givers != 0 && __popcll(idle) >= (__popcll(LM(can_give && flat)) != 0 ? QR_POOL_FLAT_MIN_IDLE : QR_POOL_MIN_IDLE))
        {
#else
        const bool can_give = pos != 0 && give > pos && (iend - give) >= QR_POOL_MIN_BYTES;
        const lm_t givers = LM(can_give);
        const lm_t idle = ~work;
        if (givers != 0 && __popcll(idle) >= QR_POOL_MIN_IDLE)
        {
#endif
            const int n_g = __popcll(givers), n_i = __popcll(idle);
            const int rank_g = lanes_below(givers), rank_i = lanes_below(idle);
            if (can_give && rank_g < n_i)
            {
#if QR_POOL_FLAT
                P.give[rank_g] = u32x4{(u32)owner, give, iend, flat ? 1u : 0u};
#else
                P.give[rank_g] = u32x4{(u32)owner, give, iend, 0u};
#endif
                iend = give; give = 0;
            }
            __syncthreads();
            const bool tk = lane_of(idle) && rank_i < n_g;
            const u32x4 g = P.give[tk ? rank_i : 0];
            const int src = tk ? (int)g.x : lane;
            /* the ray of `src` from its owner's registers (every lane of the wave executes this) */
            const float ox = __shfl(r.org.x, src), oy = __shfl(r.org.y, src), oz = __shfl(r.org.z, src);
            const float dx = __shfl(r.dir.x, src), dy = __shfl(r.dir.y, src), dz = __shfl(r.dir.z, src);
            const float tmn = __shfl(r.tmin, src), tmx = __shfl(r.tmax, src);
            const u32 osf = (u32)__shfl((int)r.osrf, src); const int ofl = __shfl(r.oflg, src);
            const float px = __shfl(r.ploc.x, src), py = __shfl(r.ploc.y, src), pz = __shfl(r.ploc.z, src);
            if (tk)
            {
                owner = src; pos = g.y; iend = g.z; give = 0; busy = true;
#if QR_POOL_FLAT
                flat = g.w != 0u;
#endif
                QR_G(g_prev = g.z; g_how = 8;)
                cr.org = {ox, oy, oz}; cr.dir = {dx, dy, dz}; cr.tmin = tmn; cr.tmax = tmx;
                cr.osrf = osf; cr.oflg = ofl; cr.ploc = {px, py, pz};
                dd = dx * dx + dy * dy + dz * dz;
                dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
                dde = dd * 1e-5f;
                w.tbuf = tmx; w.resume = 0;
                lh.srf = 0;
                cull_t = SHADOW ? tmx : u2f((u32)(P.key[src] >> 32));
                w.tbd = cull_t * dd;
            }
            __syncthreads();                    /* slots are free again */
#ifdef QR_STATS
            st_give += (unsigned long long)(n_g < n_i ? n_g : n_i);
#endif
            continue;
        }

        if (adv != 0 && __popcll(pend) < QR_DIV_BATCH)
        {
            /* ---- STEP (as walk_div) ---- */
#ifdef QR_STATS
            st_iter++; st_lanes += __popcll(adv);
#endif
            if (lane_of(adv))
            {
                QR_GUARD_POS(3, pos, ((unsigned long long)g_prev << 32) | g_how, return);
                const u32x4 a0 = *(const QR_CONST u32x4 *)(B + pos), a1 = *(const QR_CONST u32x4 *)(B + pos + 16),
                            b0 = *(const QR_CONST u32x4 *)(B + pos + 32), b1 = *(const QR_CONST u32x4 *)(B + pos + 48);
                const u32 op = a0.x, srf_off = a0.y;
                u32 next = pos + 32;
                bool stop = op == 0;
                if (SHADOW) stop = stop || ((const volatile u32 *)&P.key[owner])[0] != 0u;     /* somebody found the ray occluded */
                if (stop) next = 0;
                else
                {
                    const bool is_bv = (op & QR_OPT_BV) != 0;
                    /* the array's end in a register of its own: hipcc 7.2 otherwise reuses a0.z's register for temporaries of
                     * the exact test below and a lane that fails that test continues at a garbage offset (seen in the ISA,
                     * caught by the QR_GUARD build) */
                    u32 bv_end = a0.z;
#ifndef QR_NO_BVEND_COPY        /* -DQR_NO_BVEND_COPY: the build without the copy, for the ISA comparison in DESIGN.md */
                    asm volatile("" : "+v"(bv_end));
#endif
                    float b2, m, rhs;
                    const bool culled = (op & QR_OPF_BOX) == 0
                                     && pool_cull(srf_off, a1, is_bv ? u2f(b1.w) : u2f(a0.z), u2f(a0.w), is_bv, cr, dd, dde, dlen, w.tbd, b2, m, rhs);
                    if (is_bv)
                    {
                        next = pos + 64;
                        bool enter = !culled;
                        if (enter && !((op & QR_OPF_SPHBV) != 0 && b2 - m >= rhs))
                        {
                            /* AR_ptr itself: not a plain sphere, or the ray passes within rounding of its surface */
                            QR_FLOPS(3 + 25);
                            V3 df;
                            df.x = cr.org.x - u2f(b0.x); df.y = cr.org.y - u2f(b0.y); df.z = cr.org.z - u2f(b0.z);
                            enter = bv_hit(cr.dir, df, u2f(b1.x), u2f(b1.y), u2f(b1.z), u2f(b1.w));
                        }
                        if (!enter) next = bv_end;
                        else if (give <= pos) give = b0.w;          /* entered: a later child of this volume starts there */
                    }
                    else if (!culled) { p_op = op; p_srf = srf_off; p_pos = pos; }
                    else if ((b0.x & QR_OPT_SOLVER) != 0 && pos + 32 < iend)
                    {
                        /* second cell of the load */
                        next = pos + 64;
                        if ((b0.x & QR_OPF_BOX) != 0 || !pool_cull(b0.y, b1, u2f(b0.z), u2f(b0.w), false, cr, dd, dde, dlen, w.tbd, b2, m, rhs))
                        { p_op = b0.x; p_srf = b0.y; p_pos = pos + 32; }
                    }
                }
                if (next >= iend) next = 0;
                QR_G(g_prev = pos; g_how = (op & 0xFFFFFu) | ((next == a0.z ? 1u : 0u) << 24) | ((next == pos + 64 ? 1u : 0u) << 25);)
                pos = next;
            }
        }
        else
        {
            /* ---- SOLVE ---- */
#ifdef QR_STATS
            st_solve++; st_slanes += __popcll(pend);
#endif
            if (lane_of(pend))
            {
                SurfS s;
                ld_surf_lane(B, p_srf, s);
                const float tb = w.tbuf;
                solve_cell<SHADOW, true, true>(B, p_op, p_srf, s, cr, dd, w, lh);
                p_op = 0;
                if (SHADOW)
                {
                    if (w.resume == 0xFFFFFFFFu) { pos = 0; ((volatile u32 *)&P.key[owner])[0] = 1u; }
                }
                else if (w.tbuf != tb)
                {
                    lh_pos = p_pos;
                    w.tbd = __builtin_fminf(w.tbd, cull_t * dd);
                }
            }
        }
    }
    __syncthreads();
    if (active)
    {
        const unsigned long long k = P.key[lane];
        if (SHADOW) { if (k != 0ull) occluded = true; }
        else if ((u32)k != 0xFFFFFFFFu)
        {
            const u32x4 q = P.hit[lane];
            const u32 ss = q.x;
            h.t = u2f((u32)(k >> 32)); h.srf = ss & ~1u; h.side = (int)(ss & 1u);
            h.loc = {u2f(q.y), u2f(q.z), u2f(q.w)};
        }
    }
    __syncthreads();                            /* the next walk starts by writing the slots again */
#ifdef QR_STATS
    const unsigned long long st_start = (unsigned long long)__popcll(__ballot(active));
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        if (SHADOW) { atomicAdd(&stats[9], 1ull); atomicAdd(&stats[10], st_iter); atomicAdd(&stats[11], st_lanes);
                      atomicAdd(&stats[15], st_start); atomicAdd(&stats[22], st_solve); }
        else        { atomicAdd(&stats[16], 1ull); atomicAdd(&stats[17], st_iter); atomicAdd(&stats[18], st_lanes);
                      atomicAdd(&stats[19], st_solve); atomicAdd(&stats[20], st_slanes);
                      atomicAdd(&stats[21], st_start); }
        atomicAdd(&stats[23], st_give);
    }
#endif
}


/* ------------------------------------------------------------------------------------------------------------ */
/* walk_dda: nearest hit through the uniform grid of a long world-space list (CDda, qr_program.h)               */
/* ------------------------------------------------------------------------------------------------------------ */
#ifndef QR_DDA
#define QR_DDA 1
#endif
#ifndef QR_DDA_KMAX
#define QR_DDA_KMAX 64      /* segments per ray at most (4: +4 %, 16: +0.4 % frame time on the 10k scene) */
#endif
#ifndef QR_DDA_BATCH
#define QR_DDA_BATCH 6      /* solve as soon as this many lanes hold a candidate: an early hit ends the march of every
                             * segment behind it (16: +5 % frame time; 4 until stretches were re-split while the walk runs: with more
                             * lanes marching 6-8 are 0.6 % faster, 3 is 0.5 % slower) */
#endif
#ifndef QR_DDA_TWO_REFS
#define QR_DDA_TWO_REFS 1   /* a step looks at two refs of the current cell when the first is culled (0: one, A/B) */
#endif
#ifndef QR_DDA_SPLIT_MAXOWN
#define QR_DDA_SPLIT_MAXOWN 32
#endif
#ifndef QR_DDA_RESPLIT
#define QR_DDA_RESPLIT 1        /* stretches are halved again while the walk runs, whenever enough lanes idle (0: only the split at the start) */
#endif
#ifndef QR_DDA_RESPLIT_IDLE
#define QR_DDA_RESPLIT_IDLE 12     /* (4: -1.6 %, 8: -0.6 %, 32: -3 % against 12 on config 5) */
#endif
#ifndef QR_DDA_RESPLIT_CELLS
#define QR_DDA_RESPLIT_CELLS 2  /* a lane only gives away the far half of a stretch of at least this many cells */
#endif
#ifndef QR_DDA_SPLIT
#define QR_DDA_SPLIT 1    /* rays of a sparse wave are cut into segments marched by idle lanes */
#endif

/* bit pattern of the next float above a positive finite t (equal depths: an earlier cell of the list still wins) */
__device__ __forceinline__ float next_up(float t) { return u2f(f2u(t) + 1u); }

/*
 * Every lane walks ITS ray through the grid of ITS list: first the members too large for the grid (refs [0, n_out)),
 * then cell by cell along the ray (3D-DDA), testing the cell's refs -- copies of the members' list cells -- with the
 * conservative sphere test and, batched over the wave like walk_div's SOLVE, the solver + clip of the reference.
 * The walk of a ray ends when its depth bound lies in front of the face through which it would leave the current
 * cell.  Hits of equal depth: the reference keeps the first in list order (strict compare, tracer.cpp:1626); here
 * members are met in grid order, so a candidate whose original cell lies in front of the best hit's is tested against
 * the next float above the bound.  Same results as the list walk as long as the list's bounding volumes hold their
 * members (they only skip work); tests/test_synth.py compares against the oracle, which walks the list.
 * Sparse waves: when at most half of the lanes have a ray for the grid, every ray's interval of t inside the grid is cut
 * into 64 / rays segments and the idle lanes march the later ones (the ray from its owner's registers through
 * ds_bpermute).  A segment ends at its far end, or as soon as the best hit known for the ray -- its own or, through
 * LDS, another segment's -- lies in front of the face it would cross next.  Accepted hits meet in LDS as the minimum of
 * (depth, original cell), as in walk_pool.  Marching a later segment is wasted when an earlier one finds a hit, but it
 * occupies lanes that had nothing to do: 11 -> 30 of 64 lanes stepping, and with solver rounds from 4 candidates on
 * an early hit ends the march of everything behind it.
 */
__device__ __forceinline__ void walk_dda(BaseP B, bool active, const Ray &r, Hit &h
#ifdef QR_STATS
                                         , unsigned long long *stats
#endif
                                         )
{
    PoolLds &P = pool_lds();
    const int lane = (int)(threadIdx.x & 63u);
    Ray cr = r;                                 /* the ray this lane marches for (its own to begin with) */
    int owner = lane;
    WalkState w;
    w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0};
    w.tbuf = cr.tmax; w.resume = 0;
    float dd = cr.dir.x * cr.dir.x + cr.dir.y * cr.dir.y + cr.dir.z * cr.dir.z;
    float dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
    float dde = dd * 1e-5f;
    w.tbd = w.tbuf * dd;
    u32 best_pos = 0xFFFFFFFFu;                 /* original cell of the best hit */
    u32 last0 = 0, last1 = 0;                   /* surfaces solved last: a member spans several cells */
    Hit lh; lh.t = 0.0f; lh.srf = 0; lh.side = 0; lh.loc = {0, 0, 0};

    /* the record in front of the list program */
    u32 lo = active ? (r.list & ~31u) : 0u;
    u32x4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
    if (active)
    {
        g0 = *(const QR_CONST u32x4 *)(B + (lo - 64u)); g1 = *(const QR_CONST u32x4 *)(B + (lo - 48u));
        g2 = *(const QR_CONST u32x4 *)(B + (lo - 32u)); g3 = *(const QR_CONST u32x4 *)(B + (lo - 16u));
    }
    u32 rp = 0, rend = active ? g3.x : 0u;      /* ref cursor: the up-front members first */
    u32 p_op = 0, p_srf = 0, p_pos = 0;
    /* DDA state */
    float tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0, t_end = 0;
    float t_far = 0;                            /* finite far end of this lane's stretch: its segment's end or the grid's exit */
    int ix = 0, iy = 0, iz = 0;
    bool march = false;                         /* in the grid (pass 1) */
    /* first cell and DDA increments of a stretch of the current ray `cr` that starts at t0 */
    auto enter_at = [&](float t0) {
        const int nx = (int)(g0.w & 255u), ny = (int)((g0.w >> 8) & 255u), nz = (int)((g0.w >> 16) & 255u);
        const float ox = cr.org.x - u2f(g0.x), oy = cr.org.y - u2f(g0.y), oz = cr.org.z - u2f(g0.z);
        const float rx = __builtin_amdgcn_rcpf(cr.dir.x), ry = __builtin_amdgcn_rcpf(cr.dir.y), rz = __builtin_amdgcn_rcpf(cr.dir.z);
        const float px = ox + cr.dir.x * t0, py = oy + cr.dir.y * t0, pz = oz + cr.dir.z * t0;
        ix = cvt_floor(px * u2f(g1.x)); iy = cvt_floor(py * u2f(g1.y)); iz = cvt_floor(pz * u2f(g1.z));
        ix = ix < 0 ? 0 : (ix >= nx ? nx - 1 : ix); iy = iy < 0 ? 0 : (iy >= ny ? ny - 1 : iy); iz = iz < 0 ? 0 : (iz >= nz ? nz - 1 : iz);
        const float inf = __builtin_inff();
        tdx = cr.dir.x == 0.0f ? inf : u2f(g2.x) * __builtin_fabsf(rx);
        tdy = cr.dir.y == 0.0f ? inf : u2f(g2.y) * __builtin_fabsf(ry);
        tdz = cr.dir.z == 0.0f ? inf : u2f(g2.z) * __builtin_fabsf(rz);
        tmx = cr.dir.x == 0.0f ? inf : ((float)(ix + (cr.dir.x > 0.0f ? 1 : 0)) * u2f(g2.x) - ox) * rx;
        tmy = cr.dir.y == 0.0f ? inf : ((float)(iy + (cr.dir.y > 0.0f ? 1 : 0)) * u2f(g2.y) - oy) * ry;
        tmz = cr.dir.z == 0.0f ? inf : ((float)(iz + (cr.dir.z > 0.0f ? 1 : 0)) * u2f(g2.z) - oz) * rz;
        const u32 ci = g1.w + (u32)((iz * ny + iy) * nx + ix) * 4u;
        rp = *(const QR_CONST u32 *)(B + ci); rend = *(const QR_CONST u32 *)(B + ci + 4u);
    };
    P.key[lane] = ((unsigned long long)f2u(r.tmax) << 32) | 0xFFFFFFFFull;
    __syncthreads();
#ifdef QR_STATS
    unsigned long long st_iter = 0, st_lanes = 0, st_solve = 0, st_slanes = 0;
#endif
#pragma nounroll
    for (int pass = 0; pass < 2; pass++)
    {
        if (pass == 1)
        {
            /* ---- enter the grid: clip every ray against its box; the rays of a sparse wave are cut into 64 / rays
             *      segments of t, the later ones marched by lanes that have nothing to do ---- */
            float t_in = 0.0f, t_out = w.tbuf;
            bool enter = active;
            {
                const float ox = cr.org.x - u2f(g0.x), oy = cr.org.y - u2f(g0.y), oz = cr.org.z - u2f(g0.z);
                const float ex = u2f(g2.x) * (float)(g0.w & 255u), ey = u2f(g2.y) * (float)((g0.w >> 8) & 255u), ez = u2f(g2.z) * (float)((g0.w >> 16) & 255u);
                const float rx = __builtin_amdgcn_rcpf(cr.dir.x), ry = __builtin_amdgcn_rcpf(cr.dir.y), rz = __builtin_amdgcn_rcpf(cr.dir.z);
                const float ax = -ox * rx, bx = (ex - ox) * rx, ay = -oy * ry, by = (ey - oy) * ry, az = -oz * rz, bz = (ez - oz) * rz;
                if (cr.dir.x == 0.0f) enter = enter && !(ox < 0.0f || ox > ex); else { t_in = __builtin_fmaxf(t_in, __builtin_fminf(ax, bx)); t_out = __builtin_fminf(t_out, __builtin_fmaxf(ax, bx)); }
                if (cr.dir.y == 0.0f) enter = enter && !(oy < 0.0f || oy > ey); else { t_in = __builtin_fmaxf(t_in, __builtin_fminf(ay, by)); t_out = __builtin_fminf(t_out, __builtin_fmaxf(ay, by)); }
                if (cr.dir.z == 0.0f) enter = enter && !(oz < 0.0f || oz > ez); else { t_in = __builtin_fmaxf(t_in, __builtin_fminf(az, bz)); t_out = __builtin_fminf(t_out, __builtin_fmaxf(az, bz)); }
                enter = enter && (t_in <= t_out);
            }
            const lm_t owners = LM(enter);
            const int n_own = __popcll(owners);
            if (n_own == 0) break;
            int K = 64 / n_own; K = K > QR_DDA_KMAX ? QR_DDA_KMAX : K;
            if (n_own > QR_DDA_SPLIT_MAXOWN) K = 1;
            int seg = 0;
            march = enter;
#if QR_DDA_SPLIT
            if (K > 1)
            {
                const lm_t idle = ~owners;
                const int rank_o = lanes_below(owners), rank_i = lanes_below(idle);
                if (enter) P.give[rank_o] = u32x4{(u32)lane, 0u, 0u, 0u};
                __syncthreads();
                const bool help = !enter && rank_i < n_own * (K - 1);
                const int src = help ? (int)P.give[rank_i % n_own].x : lane;
                if (help) seg = 1 + rank_i / n_own;
                /* the ray and what its owner knows so far, from the owner's registers (all lanes execute this) */
                const float f_ox = __shfl(cr.org.x, src), f_oy = __shfl(cr.org.y, src), f_oz = __shfl(cr.org.z, src);
                const float f_dx = __shfl(cr.dir.x, src), f_dy = __shfl(cr.dir.y, src), f_dz = __shfl(cr.dir.z, src);
                const float f_tmn = __shfl(cr.tmin, src), f_tb = __shfl(w.tbuf, src);
                const u32 f_osf = (u32)__shfl((int)cr.osrf, src); const int f_ofl = __shfl(cr.oflg, src);
                const float f_px = __shfl(cr.ploc.x, src), f_py = __shfl(cr.ploc.y, src), f_pz = __shfl(cr.ploc.z, src);
                const u32 f_lo = (u32)__shfl((int)lo, src), f_bp = (u32)__shfl((int)best_pos, src);
                const float f_ti = __shfl(t_in, src), f_to = __shfl(t_out, src);
                if (help)
                {
                    owner = src; march = true;
                    cr.org = {f_ox, f_oy, f_oz}; cr.dir = {f_dx, f_dy, f_dz}; cr.tmin = f_tmn; cr.tmax = f_tb;
                    cr.osrf = f_osf; cr.oflg = f_ofl; cr.ploc = {f_px, f_py, f_pz};
                    dd = f_dx * f_dx + f_dy * f_dy + f_dz * f_dz;
                    dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
                    dde = dd * 1e-5f;
                    w.tbuf = f_tb; w.tbd = f_tb * dd; w.resume = 0;
                    best_pos = f_bp; lh.srf = 0; last0 = 0; last1 = 0;
                    lo = f_lo; t_in = f_ti; t_out = f_to;
                    g0 = *(const QR_CONST u32x4 *)(B + (lo - 64u)); g1 = *(const QR_CONST u32x4 *)(B + (lo - 48u));
                    g2 = *(const QR_CONST u32x4 *)(B + (lo - 32u)); g3 = *(const QR_CONST u32x4 *)(B + (lo - 16u));
                }
                __syncthreads();
            }
            else K = 1;
#else
            K = 1;
#endif
            if (march)
            {
                /* this lane's segment of t, its first cell and the DDA increments */
                const float dt = (t_out - t_in) * (1.0f / (float)K);
                const float t0 = seg == 0 ? t_in : t_in + dt * (float)seg;
                t_end = seg == K - 1 ? __builtin_inff() : t_in + dt * (float)(seg + 1);
                t_far = seg == K - 1 ? t_out : t_end;
                enter_at(t0);
            }
        }
        for (;;)
        {
            const lm_t pend = LM(p_op != 0);
            const lm_t adv = (pass == 0 ? LM(rp < rend) : LM(march)) & ~pend;
            if ((adv | pend) == 0) break;
#if QR_DDA_RESPLIT
            if (pass == 1)
            {
                /* Stretches end at very different times (a hit right behind the start, or a march through the whole grid): as
                 * soon as QR_DDA_RESPLIT_IDLE lanes have nothing to march, lanes with a long stretch ahead hand its far half to
                 * them -- the walk lasts as long as its longest stretch, and that is halved.  The same split as the one at the
                 * start (the cell around the cut is looked at from both sides, hits meet in LDS as the minimum of depth and
                 * original cell), only later. */
                const lm_t idle = ~(LM(march) | pend);
                const float cur_t = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
                const float lim = __builtin_fminf(t_far, __builtin_fminf(w.tbuf, u2f(((const volatile u32 *)&P.key[owner])[1])));
                const bool can_give = march && p_op == 0 && (lim - cur_t) > (float)QR_DDA_RESPLIT_CELLS * __builtin_fminf(tdx, __builtin_fminf(tdy, tdz));
                const lm_t givers = LM(can_give);
                if (givers != 0 && __popcll(idle) >= QR_DDA_RESPLIT_IDLE)
                {
                    const int n_g = __popcll(givers), n_i = __popcll(idle);
                    const int rank_g = lanes_below(givers), rank_i = lanes_below(idle);
                    if (can_give && rank_g < n_i)
                    {
                        const float mid = 0.5f * (cur_t + lim);
                        P.give[rank_g] = u32x4{(u32)lane, f2u(mid), f2u(t_end), f2u(t_far)};
                        t_end = mid; t_far = mid;
                    }
                    __syncthreads();
                    const bool tk = lane_of(idle) && rank_i < n_g;
                    const u32x4 gv = P.give[tk ? rank_i : 0];
                    const int src = tk ? (int)gv.x : lane;
                    const float f_ox = __shfl(cr.org.x, src), f_oy = __shfl(cr.org.y, src), f_oz = __shfl(cr.org.z, src);
                    const float f_dx = __shfl(cr.dir.x, src), f_dy = __shfl(cr.dir.y, src), f_dz = __shfl(cr.dir.z, src);
                    const float f_tmn = __shfl(cr.tmin, src), f_tb = __shfl(w.tbuf, src);
                    const u32 f_osf = (u32)__shfl((int)cr.osrf, src); const int f_ofl = __shfl(cr.oflg, src);
                    const float f_px = __shfl(cr.ploc.x, src), f_py = __shfl(cr.ploc.y, src), f_pz = __shfl(cr.ploc.z, src);
                    const u32 f_lo = (u32)__shfl((int)lo, src), f_bp = (u32)__shfl((int)best_pos, src);
                    const int f_own = __shfl(owner, src);
                    if (tk)
                    {
                        owner = f_own; march = true;
                        cr.org = {f_ox, f_oy, f_oz}; cr.dir = {f_dx, f_dy, f_dz}; cr.tmin = f_tmn; cr.tmax = f_tb;
                        cr.osrf = f_osf; cr.oflg = f_ofl; cr.ploc = {f_px, f_py, f_pz};
                        dd = f_dx * f_dx + f_dy * f_dy + f_dz * f_dz;
                        dlen = __builtin_amdgcn_sqrtf(dd) * 1.000001f;
                        dde = dd * 1e-5f;
                        w.tbuf = f_tb; w.tbd = f_tb * dd; w.resume = 0;
                        best_pos = f_bp; lh.srf = 0; last0 = 0; last1 = 0;
                        lo = f_lo;
                        g0 = *(const QR_CONST u32x4 *)(B + (lo - 64u)); g1 = *(const QR_CONST u32x4 *)(B + (lo - 48u));
                        g2 = *(const QR_CONST u32x4 *)(B + (lo - 32u)); g3 = *(const QR_CONST u32x4 *)(B + (lo - 16u));
                        t_end = u2f(gv.z); t_far = u2f(gv.w);
                        enter_at(u2f(gv.y));
                    }
                    __syncthreads();
                    continue;
                }
            }
#endif
            if (adv != 0 && __popcll(pend) < QR_DDA_BATCH)
            {
#ifdef QR_STATS
                st_iter++; st_lanes += __popcll(adv);
#endif
                if (lane_of(adv))
                {
                    if (rp < rend)
                    {
                        /* ---- next ref of the current cell (or of the up-front members) ---- */
                        const u32 ro = g2.w + rp * 32u;
                        const u32x4 a0 = *(const QR_CONST u32x4 *)(B + ro), a1 = *(const QR_CONST u32x4 *)(B + ro + 16);
#if QR_DDA_TWO_REFS
                        /* the ref behind it travels in the same round trip (the array has two refs of slack) and is looked at in
                         * this step when the first one is culled: the march is a chain of dependent loads, one per step */
                        const u32x4 c0 = *(const QR_CONST u32x4 *)(B + ro + 32), c1 = *(const QR_CONST u32x4 *)(B + ro + 48);
#endif
                        rp++;
                        if (a0.y != last0 && a0.y != last1)
                        {
                            float b2, m, rhs;
                            const float r2 = u2f(a0.z);
                            if (!pool_cull(a0.y, a1, r2, r2 * 1.01f, false, cr, dd, dde, dlen, w.tbd, b2, m, rhs))
                            { p_op = a0.x; p_srf = a0.y; p_pos = a0.w; }
                        }
#if QR_DDA_TWO_REFS
                        if (p_op == 0 && rp < rend)
                        {
                            rp++;
                            if (c0.y != last0 && c0.y != last1)
                            {
                                float b2, m, rhs;
                                const float r2 = u2f(c0.z);
                                if (!pool_cull(c0.y, c1, r2, r2 * 1.01f, false, cr, dd, dde, dlen, w.tbd, b2, m, rhs))
                                { p_op = c0.x; p_srf = c0.y; p_pos = c0.w; }
                            }
                        }
#endif
                    }
                    else
                    {
                        /* ---- leave the cell through its nearest face, unless the best hit known for the ray -- this
                         *      lane's or another segment's -- lies in front of that face, or the segment ends there ---- */
                        const float t_exit = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
                        const float known = __builtin_fminf(w.tbuf, u2f(((const volatile u32 *)&P.key[owner])[1]));
                        if (known < t_exit || t_exit >= t_end) march = false;
                        else
                        {
                            const int nx = (int)(g0.w & 255u), ny = (int)((g0.w >> 8) & 255u), nz = (int)((g0.w >> 16) & 255u);
                            if (tmx <= tmy && tmx <= tmz) { ix += cr.dir.x > 0.0f ? 1 : -1; tmx += tdx; }
                            else if (tmy <= tmz)          { iy += cr.dir.y > 0.0f ? 1 : -1; tmy += tdy; }
                            else                          { iz += cr.dir.z > 0.0f ? 1 : -1; tmz += tdz; }
                            if ((unsigned)ix >= (unsigned)nx || (unsigned)iy >= (unsigned)ny || (unsigned)iz >= (unsigned)nz) march = false;
                            else
                            {
                                const u32 ci = g1.w + (u32)((iz * ny + iy) * nx + ix) * 4u;
                                rp = *(const QR_CONST u32 *)(B + ci); rend = *(const QR_CONST u32 *)(B + ci + 4u);
                            }
                        }
                    }
                }
            }
            else
            {
                /* ---- SOLVE ---- */
#ifdef QR_STATS
                st_solve++; st_slanes += __popcll(pend);
#endif
                bool accepted = false;
                if (lane_of(pend))
                {
                    SurfS s;
                    ld_surf_lane(B, p_srf, s);
                    const float tb = w.tbuf;
                    const float tt = (p_pos < best_pos && best_pos != 0xFFFFFFFFu) ? next_up(tb) : tb;
                    w.tbuf = tt;
                    solve_cell<false, true, true>(B, p_op, p_srf, s, cr, dd, w, lh);
                    if (w.tbuf != tt) { best_pos = p_pos; accepted = true; }    /* solve_cell stored the new bound */
                    else w.tbuf = tb;
                    last1 = last0; last0 = p_srf;
                    p_op = 0;
                }
                /* hits meet in LDS per ray: minimum of (depth, original cell) over all segments */
                if (any_lane(accepted))
                {
                    const unsigned long long k = ((unsigned long long)f2u(lh.t) << 32) | (unsigned long long)best_pos;
                    if (accepted) atomicMin(&P.key[owner], k);
                    __syncthreads();
                    if (accepted && P.key[owner] == k)
                        P.hit[owner] = u32x4{lh.srf | (u32)lh.side, f2u(lh.loc.x), f2u(lh.loc.y), f2u(lh.loc.z)};
                }
            }
        }
    }
    __syncthreads();
    if (active)
    {
        const unsigned long long k = P.key[lane];
        if ((u32)k != 0xFFFFFFFFu)
        {
            const u32x4 q = P.hit[lane];
            h.t = u2f((u32)(k >> 32)); h.srf = q.x & ~1u; h.side = (int)(q.x & 1u);
            h.loc = {u2f(q.y), u2f(q.z), u2f(q.w)};
        }
    }
    __syncthreads();
#ifdef QR_STATS
    const unsigned long long st_start = (unsigned long long)__popcll(__ballot(active));
    if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63))
    {
        atomicAdd(&stats[16], 1ull); atomicAdd(&stats[17], st_iter); atomicAdd(&stats[18], st_lanes);
        atomicAdd(&stats[19], st_solve); atomicAdd(&stats[20], st_slanes); atomicAdd(&stats[21], st_start);
    }
#endif
}

/*
 * Wave-wide traversal: lanes with `active` walk their lists.  Lanes that share a list head are walked together
 * (wave-packet walk) as long as the wave is coherent; when the leading group is a small part of what is left
 * (many different lists: secondary hits on many small objects), or the list is a long hierarchy on which rays
 * part ways (QR_LISTF_LONG) and the caller does not vouch for the rays' coherence (`coherent`: primary rays and
 * the shadow rays of primary hits -- neighbouring pixels, same light), the remaining lanes whose lists allow it
 * walk per lane.  The low bits of a list offset carry these flags (cells are 32-byte aligned).
 * DIVK = false is the kernel instance for scenes without long hierarchies (every scene the reference engine
 * prepares): the per-lane walk is compiled out there, which is worth 4 % of instructions through register pressure.
 */
template <bool SHADOW, bool DIVK>
__device__ __forceinline__ void traverse(BaseP B, bool active, bool coherent, const Ray &r, Hit &h, bool &occluded
#ifdef QR_STATS
                                         , unsigned long long *stats
#endif
                                         )
{
    h.t = r.tmax; h.srf = 0; h.side = 0; h.loc = {0, 0, 0};
    occluded = false;
    lm_t pending = LM(active && r.list != 0);
    while (pending != 0)
    {
        const int leader = __ffsll((long long)pending) - 1;
        const u32 head = (u32)__builtin_amdgcn_readlane((int)r.list, leader);
        const lm_t mine = pending & LM(r.list == head);
#ifdef QR_WAVETIME
        if ((int)(threadIdx.x & 63u) == leader) qr_wt_groups[SHADOW ? 1 : 0]++;
#endif
        if constexpr (DIVK)
        {
        const int n_left = __popcll(pending), n_mine = __popcll(mine);
        const lm_t can_div = pending & LM((r.list & QR_LISTF_DIV) != 0);
        const bool incoherent = n_mine * QR_INCOH_RATIO <= n_left * QR_INCOH_DEN && n_left >= QR_INCOH_MIN;
        const bool long_list = !coherent && (head & QR_LISTF_LONG) != 0 && (head & QR_LISTF_DIV) != 0 && n_mine >= QR_LONG_MIN;
        if ((incoherent || long_list) && can_div != 0)
        {
            const lm_t go = incoherent ? can_div : mine;
            pending &= ~go;
#if QR_DDA
            /* nearest-hit rays on lists that carry a uniform grid */
            if (!SHADOW && (go & LM((r.list & QR_LISTF_DDA) == 0)) == 0)
            {
                walk_dda(B, lane_of(go), r, h
#ifdef QR_STATS
                         , stats
#endif
                         );
                continue;
            }
#endif
#if QR_POOL
            /* hand-over needs lists without transform state (QR_LISTF_WORLD) */
            if ((go & LM((r.list & QR_LISTF_WORLD) == 0)) == 0)
            walk_pool<SHADOW>(B, lane_of(go), r, h, occluded
#ifdef QR_STATS
                              , stats
#endif
                              );
            else
#endif
            walk_div<SHADOW>(B, lane_of(go), r, h, occluded
#ifdef QR_STATS
                             , stats
#endif
                             );
            continue;
        }
        }
        pending &= ~mine;
        if (lane_of(mine))
        {
            walk_list<SHADOW, !DIVK>(B, head & ~31u, r, h, occluded
#ifdef QR_STATS
                              , stats
#endif
                              );
        }
    }
}

#endif /* QR_WALK_HPP */
