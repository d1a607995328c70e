/*
 * qr_bounds.hpp - host side: conservative world-space bounds of a surface's visible part, derived from the
 * snapshot alone.  Used for the kernel's wave-level cull (bounding sphere in every list cell) and by the tile
 * binning pass (sphere interval + clipped bounding box).  Included by qr_device.hip only.
 */
#ifndef QR_BOUNDS_HPP
#define QR_BOUNDS_HPP

/*
 * Conservative WORLD-space bounding sphere of the visible part of surface `i`, derived from
 * the snapshot only (clip box, shape coefficients, transform).  r = +inf when no bound can be
 * shown.  Used by the list walk purely as a wave-level cull: an element is skipped when every
 * ray of the group provably misses the sphere (QR_CULL in qr_kernel.hpp), so a wrong "inf" costs
 * time, never correctness; the radius is inflated so that fp32 rounding in the device test and
 * in the reference's hit points cannot turn a real hit into a cull.
 */
struct BBox { bool valid; double p[8][3]; };     /* world-space corners of the (oriented) bounding box */

static BSphere bound_sphere(const qr_scene_view &v, int i, BBox *box = nullptr)
{
    const double INF = 1e300;
    BSphere out;
    out.c[0] = out.c[1] = out.c[2] = 0.0f; out.r = __builtin_inff();
    BBox own_box;
    if (box == nullptr) box = &own_box;          /* the corners also give the axis-aligned box */
    box->valid = false;
    const qr_surface &q = v.srf[i];
    if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) return out;
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++)
    {
        lo[k] = (q.minmax_t & (1u << k)) ? (double)q.min[k] : -INF;
        hi[k] = (q.minmax_t & (1u << (3 + k))) ? (double)q.max[k] : INF;
        if (lo[k] > hi[k]) { lo[k] = hi[k] = 0.5 * (lo[k] + hi[k]); }    /* empty box: nothing visible */
    }
    /*
     * Custom clippers that bound the surface (engine.cpp:1845-1947 builds the list, tracer.cpp:1931-2151
     * applies it): a clipper kept on its inner side (data > 0: f <= 0) that is a closed quadric confines the
     * surface to that quadric's box; a plane clipper to a half-space.  Only untransformed pairs outside
     * accumulator segments are used -- ignoring a clipper is always conservative.
     */
    if (q.has_trm == 0 && q.clip != QR_NULL)
    {
        bool plain = true;
        int guard = 0;
        for (int e = q.clip; e != QR_NULL && plain; e = v.elm[e].next)
            if (v.elm[e].simd == QR_NULL || ++guard > 4096) plain = false;
        for (int e = q.clip; e != QR_NULL && plain; e = v.elm[e].next)
        {
            const qr_elem &ce = v.elm[e];
            if (ce.simd < 0 || ce.simd >= (int)v.hdr->n_srf) continue;
            const qr_surface &c = v.srf[ce.simd];
            if (c.has_trm != 0 || c.srf_t[3] < 0 || c.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
            if (c.srf_t[2] == 1)
            {
                /* PL_clp 4198-4208: f = +-(x_a - pos_a); APPLY_CLIP: data < 0 keeps f >= 0, data > 0 keeps f <= 0 */
                const int a = (int)((c.axes >> 4) & 3);
                if (a > 2) continue;
                const bool neg = ((c.axes >> 10) & 1) != 0;
                const double lim = (double)c.pos[a] - (double)q.pos[a];
                const bool keep_le = (ce.data > 0) != neg;      /* x_a <= pos_a survives */
                if (keep_le) { if (hi[a] > lim + 1e-4) hi[a] = lim + 1e-4; }
                else         { if (lo[a] < lim - 1e-4) lo[a] = lim - 1e-4; }
            }
            else if ((c.srf_t[2] == 2 || c.srf_t[2] == 3) && ce.data > 0 &&
                     c.sci[0] > 0.0f && c.sci[1] > 0.0f && c.sci[2] > 0.0f && c.sci[3] > 0.0f &&
                     (c.srf_t[2] == 3 || (c.scj[0] == 0.0f && c.scj[1] == 0.0f && c.scj[2] == 0.0f)))
            {
                for (int a = 0; a < 3; a++)
                {
                    const double ext = __builtin_sqrt((double)c.sci[3] / (double)c.sci[a]) * 1.0005 + 1e-4;
                    const double cl = (double)c.pos[a] - (double)q.pos[a];
                    if (lo[a] < cl - ext) lo[a] = cl - ext;
                    if (hi[a] > cl + ext) hi[a] = cl + ext;
                }
            }
        }
        for (int k = 0; k < 3; k++) if (lo[k] > hi[k]) { lo[k] = hi[k] = 0.5 * (lo[k] + hi[k]); }
    }
    const int solver = q.srf_t[0];
    if (solver == 1)
    {
        const int k = (int)((q.axes >> 4) & 3);
        if (k > 2) return out;
        lo[k] = lo[k] > 0.0 ? lo[k] : (hi[k] < 0.0 ? hi[k] : 0.0);
        hi[k] = lo[k];
        lo[k] -= 1e-3; hi[k] += 1e-3;
    }
    else if (solver == 2 || solver == 3)
    {
        /* sum_a sci_a x_a^2 - 2 sum_a scj_a x_a = sci_w  ->  for an axis with sci_a > 0, scj_a == 0:
         * sci_a x_a^2 <= sci_w + sum_{b != a} [ max(2 scj_b x_b) + max(-sci_b x_b^2) ] over the box */
        for (int pass = 0; pass < 3; pass++)
            for (int a = 0; a < 3; a++)
            {
                const double sa = q.sci[a];
                if (sa == 0.0 && q.scj[a] != 0.0f)
                {
                    /* linear axis (paraboloids): 2 scj_a x_a = sum_{b != a} (sci_b x_b^2 - 2 scj_b x_b) - sci_w */
                    double L = -(double)q.sci[3], U = -(double)q.sci[3];
                    for (int b = 0; b < 3; b++)
                    {
                        if (b == a) continue;
                        const double sb = q.sci[b], jb = q.scj[b];
                        const double m2 = (lo[b] * lo[b] > hi[b] * hi[b]) ? lo[b] * lo[b] : hi[b] * hi[b];   /* max x^2 */
                        if (sb > 0.0) { U += (m2 < INF / 4) ? sb * m2 : INF; }
                        else if (sb < 0.0) { L += (m2 < INF / 4) ? sb * m2 : -INF; }
                        if (jb != 0.0)
                        {
                            const double e0 = -2.0 * jb * lo[b], e1 = -2.0 * jb * hi[b];
                            const double mn = e0 < e1 ? e0 : e1, mx = e0 > e1 ? e0 : e1;
                            L += (mn > -INF / 4) ? mn : -INF; U += (mx < INF / 4) ? mx : INF;
                        }
                    }
                    const double d2 = 2.0 * (double)q.scj[a];
                    double l = (d2 > 0.0 ? L : U) / d2, h = (d2 > 0.0 ? U : L) / d2;
                    if (l > -INF / 8) { l -= 1e-4 + 5e-4 * (l < 0 ? -l : l); if (lo[a] < l) lo[a] = l; }
                    if (h < INF / 8) { h += 1e-4 + 5e-4 * (h < 0 ? -h : h); if (hi[a] > h) hi[a] = h; }
                    if (lo[a] > hi[a]) lo[a] = hi[a] = 0.5 * (lo[a] + hi[a]);
                    continue;
                }
                if (!(sa > 0.0) || q.scj[a] != 0.0f) continue;
                double rhs = q.sci[3];
                bool ok = true;
                for (int b = 0; b < 3 && ok; b++)
                {
                    if (b == a) continue;
                    const double sb = q.sci[b], jb = q.scj[b];
                    if (jb != 0.0)
                    {
                        const double e0 = 2.0 * jb * lo[b], e1 = 2.0 * jb * hi[b];
                        const double m = e0 > e1 ? e0 : e1;
                        if (!(m < INF / 4)) ok = false; else rhs += m;
                    }
                    if (sb < 0.0)
                    {
                        const double x2 = (lo[b] * lo[b] > hi[b] * hi[b]) ? lo[b] * lo[b] : hi[b] * hi[b];
                        if (!(x2 < INF / 4)) ok = false; else rhs += -sb * x2;
                    }
                    /* sb >= 0: -sb x_b^2 <= 0, dropped */
                }
                if (!ok) continue;
                const double lim = __builtin_sqrt((rhs > 0.0 ? rhs : 0.0) / sa) * 1.0005 + 1e-4;
                if (lo[a] < -lim) lo[a] = -lim;
                if (hi[a] > lim) hi[a] = lim;
                if (lo[a] > hi[a]) lo[a] = hi[a] = 0.5 * (lo[a] + hi[a]);
            }
    }
    else
    {
        return out;
    }
    for (int k = 0; k < 3; k++) if (!(lo[k] > -INF / 4) || !(hi[k] < INF / 4)) return out;

    double cl[3], r2 = 0.0;
    for (int k = 0; k < 3; k++) { cl[k] = 0.5 * (lo[k] + hi[k]); const double h = 0.5 * (hi[k] - lo[k]); r2 += h * h; }
    double rl = __builtin_sqrt(r2);
    if (solver == 2 && q.scj[0] == 0.0f && q.scj[1] == 0.0f && q.scj[2] == 0.0f && q.sci[0] > 0.0f && q.sci[1] > 0.0f && q.sci[2] > 0.0f && q.sci[3] > 0.0f)
    {
        /* a closed ellipsoid about the local origin: the sphere of its largest half-axis holds it -- for a ball that is the
         * ball itself, where the box's circumsphere is sqrt(3) times too wide */
        const double smin = q.sci[0] < q.sci[1] ? (q.sci[0] < q.sci[2] ? q.sci[0] : q.sci[2]) : (q.sci[1] < q.sci[2] ? q.sci[1] : q.sci[2]);
        const double re = __builtin_sqrt((double)q.sci[3] / smin) * 1.0005 + 1e-4;
        if (re < rl) { rl = re; cl[0] = cl[1] = cl[2] = 0.0; }
    }
    double cw[3];
    if (q.has_trm == 0)
    {
        for (int k = 0; k < 3; k++) cw[k] = (double)q.pos[k] + cl[k];
        if (box)
        {
            for (int c = 0; c < 8; c++)
                for (int k = 0; k < 3; k++) box->p[c][k] = (double)q.pos[k] + (((c >> k) & 1) ? hi[k] : lo[k]);
            box->valid = true;
        }
    }
    else
    {
        if (q.trnode < 0) return out;
        const qr_surface &t = v.srf[q.trnode];
        double m[3][3] = { { t.tci[0], t.tci[1], t.tci[2] }, { t.tcj[0], t.tcj[1], t.tcj[2] }, { t.tck[0], t.tck[1], t.tck[2] } };
        if (t.has_trm == 1) { m[0][1] = m[0][2] = m[1][0] = m[1][2] = m[2][0] = m[2][1] = 0.0; }
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
                         - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        if (!(det > 1e-12 || det < -1e-12)) return out;
        double inv[3][3];
        inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det;
        inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
        inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
        inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det;
        inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det;
        inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
        inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det;
        inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
        inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
        double pl[3];       /* point in the trnode's frame */
        for (int k = 0; k < 3; k++) pl[k] = cl[k] + (q.trnode == i ? 0.0 : (double)q.pos[k]);
        double fro = 0.0;
        for (int a = 0; a < 3; a++)
        {
            cw[a] = (double)t.pos[a];
            for (int b = 0; b < 3; b++) { cw[a] += inv[a][b] * pl[b]; fro += inv[a][b] * inv[a][b]; }
        }
        rl *= __builtin_sqrt(fro);
        if (box)
        {
            for (int c = 0; c < 8; c++)
                for (int a = 0; a < 3; a++)
                {
                    double acc = (double)t.pos[a];
                    for (int b = 0; b < 3; b++)
                        acc += inv[a][b] * ((((c >> b) & 1) ? hi[b] : lo[b]) + (q.trnode == i ? 0.0 : (double)q.pos[b]));
                    box->p[c][a] = acc;
                }
            box->valid = true;
        }
    }
    const double r = rl * 1.002 + 2e-3;
    if (!(r < 1e30)) return out;
    for (int k = 0; k < 3; k++) { if (!(cw[k] > -1e30 && cw[k] < 1e30)) return out; out.c[k] = (float)cw[k]; }
    out.r = (float)r * 1.0001f + 1e-6f;
    if (box->valid)
    {
        /* axis-aligned box of the corners, inflated like the sphere (relative + absolute, then outward-rounded to float) */
        for (int k = 0; k < 3; k++)
        {
            double mn = box->p[0][k], mx = box->p[0][k];
            for (int c = 1; c < 8; c++) { if (box->p[c][k] < mn) mn = box->p[c][k]; if (box->p[c][k] > mx) mx = box->p[c][k]; }
            const double pad = 2e-3 + 1e-3 * (mx - mn) + 1e-5 * ((mn < 0 ? -mn : mn) + (mx < 0 ? -mx : mx));
            out.lo[k] = __builtin_nextafterf((float)(mn - pad), -__builtin_inff());
            out.hi[k] = __builtin_nextafterf((float)(mx + pad), __builtin_inff());
        }
    }
    return out;
}

#endif /* QR_BOUNDS_HPP */
