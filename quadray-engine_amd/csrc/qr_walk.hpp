/*
 * qr_walk.hpp - device code, list traversal: clip() (CC_clp), walk_element (one list element: diff/transform,
 * bounding volumes, plane/quadric/two-plane solvers, candidates), walk_list (wave-packet walk of one list with
 * the bounding-sphere cull and the array jump), walk_div (per-lane divergent walk), traverse (groups lanes by
 * list head).  Included by qr_kernel.hpp after the shared types.
 */
#ifndef QR_WALK_HPP
#define QR_WALK_HPP

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

#endif /* QR_WALK_HPP */
