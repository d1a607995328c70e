/*
 * qr_device.hip - GPU half of the C ABI (include/qrhip.h): scene upload,
 * launches, timing, and qr_render0 (the reference entry point replacement,
 * core/tracer/tracer.cpp:1081 / dispatcher 5992-6104).
 *
 * No CPU fallback exists: every entry point here fails with QR_ERR_DEVICE when
 * no HIP device is usable.
 */
#include "qr_internal.h"
#include "qr_kernel.hpp"

#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>
#include <string>

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess)                                                  \
            return qr_fail(QR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct qr_device_scene
{
    int device;
    void *d_blob;               /* one allocation holding every array      */
    uint64_t blob_bytes;
    DevScene sc;                /* device pointers + launch parameters      */
    qr_frame fr;                /* host copy of the frame parameters        */
    qr_header hdr;
    unsigned long long *d_counters;
    hipEvent_t ev0, ev1;
};

extern "C" const char *qr_kernel_name(void) { return "qr_render_kernel"; }

extern "C" int qr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static size_t pad16(size_t x) { return (x + 15) & ~(size_t)15; }

/*
 * Conservative WORLD-space bounding sphere of the visible part of surface `i`, derived from
 * the snapshot only (clip box, shape coefficients, transform).  r = +inf when no bound can be
 * shown.  Used by the list walk purely as a wave-level cull: an element is skipped when every
 * ray of the group provably misses the sphere (QR_CULL in qr_kernel.hpp), so a wrong "inf" costs
 * time, never correctness; the radius is inflated so that fp32 rounding in the device test and
 * in the reference's hit points cannot turn a real hit into a cull.
 */
struct BSphere { float c[3]; float r; };

static BSphere bound_sphere(const qr_scene_view &v, int i)
{
    const double INF = 1e300;
    BSphere out = { {0.0f, 0.0f, 0.0f}, __builtin_inff() };
    const qr_surface &q = v.srf[i];
    if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) return out;
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++)
    {
        lo[k] = (q.minmax_t & (1u << k)) ? (double)q.min[k] : -INF;
        hi[k] = (q.minmax_t & (1u << (3 + k))) ? (double)q.max[k] : INF;
        if (lo[k] > hi[k]) { lo[k] = hi[k] = 0.5 * (lo[k] + hi[k]); }    /* empty box: nothing visible */
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
    double cw[3];
    if (q.has_trm == 0)
    {
        for (int k = 0; k < 3; k++) cw[k] = (double)q.pos[k] + cl[k];
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
    }
    const double r = rl * 1.002 + 2e-3;
    if (!(r < 1e30)) return out;
    for (int k = 0; k < 3; k++) { if (!(cw[k] > -1e30 && cw[k] < 1e30)) return out; out.c[k] = (float)cw[k]; }
    out.r = (float)r * 1.0001f + 1e-6f;
    return out;
}

extern "C" int qr_scene_upload(const void *blob, uint64_t size, int device, qr_device_scene **out)
{
    if (blob == nullptr || out == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    qr_scene_view v;
    int rc = qr_scene_view_init(&v, blob, size);
    if (rc != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot (qr_scene_view_init " + std::to_string(rc) + ")");
    const qr_frame &fr = *v.frame;
    if (fr.fsaa < 0 || fr.fsaa > 2) return qr_fail(QR_ERR_UNSUP, "unsupported fsaa");
    if (fr.frm_w <= 0 || fr.frm_h <= 0 || fr.tile_w <= 0 || fr.tile_h <= 0) return qr_fail(QR_ERR_ARG, "bad frame parameters");

    /* host-side validation of every index the kernel will follow, so that a
     * malformed snapshot cannot turn into an out-of-bounds device access */
    const int n_srf = (int)v.hdr->n_srf, n_mat = (int)v.hdr->n_mat, n_lgt = (int)v.hdr->n_lgt;
    const int n_elm = (int)v.hdr->n_elm, n_tex = (int)v.hdr->n_texels;
    auto ok_elm = [&](int i) { return i == QR_NULL || (i >= 0 && i < n_elm); };
    auto ok_srf = [&](int i) { return i >= 0 && i < n_srf; };
    for (int i = 0; i < n_elm; i++)
    {
        const qr_elem &e = v.elm[i];
        if (!ok_elm(e.next)) return qr_fail(QR_ERR_ARG, "element next out of range");
    }
    for (uint32_t i = 0; i < v.hdr->n_tiles; i++)
        if (!ok_elm(v.tiles[i])) return qr_fail(QR_ERR_ARG, "tile head out of range");
    if (!ok_elm(fr.clist)) return qr_fail(QR_ERR_ARG, "clist out of range");
    for (int i = 0; i < n_mat; i++)
    {
        const qr_material &m = v.mat[i];
        uint64_t n = (uint64_t)(m.xmask + 1) * (m.ymask + 1);
        if (m.tex < 0 || (uint64_t)m.tex + n > (uint64_t)n_tex) return qr_fail(QR_ERR_ARG, "texture out of range");
        if (m.t_map[0] < 0 || m.t_map[0] > 1 || m.t_map[1] < 0 || m.t_map[1] > 1) return qr_fail(QR_ERR_ARG, "bad t_map");
        if ((m.xmask & (m.xmask + 1)) != 0 || (m.ymask & (m.ymask + 1)) != 0) return qr_fail(QR_ERR_ARG, "texture size not a power of two");
        if (((uint64_t)m.ymask << (m.yshft & 31)) + m.xmask >= n) return qr_fail(QR_ERR_ARG, "texture addressing exceeds texture");
    }
    /* classify lists: walk every list once with a step bound (cycle check) */
    auto check_list = [&](int head, int kind) -> const char * {
        /* kind 0 surfaces, 1 clippers, 2 lights */
        int cnt = 0;
        for (int e = head; e != QR_NULL; e = v.elm[e].next)
        {
            if (++cnt > n_elm) return "cyclic list";
            const qr_elem &el = v.elm[e];
            if (kind == 2)
            {
                if (el.simd < 0 || el.simd >= n_lgt) return "light index out of range";
                if (!ok_elm(el.data)) return "shadow list out of range";
            }
            else if (kind == 0)
            {
                if (!ok_srf(el.simd)) return "surface index out of range";
                if (el.data != QR_NULL && !ok_elm(el.data)) return "array last element out of range";
            }
            else
            {
                if (el.simd != QR_NULL)
                {
                    if (!ok_srf(el.simd)) return "clipper index out of range";
                    if (v.srf[el.simd].srf_t[3] < 0 && !ok_elm(el.data)) return "clip trnode last out of range";
                }
            }
        }
        return nullptr;
    };
    for (uint32_t i = 0; i < v.hdr->n_tiles; i++)
        if (const char *m = check_list(v.tiles[i], 0)) return qr_fail(QR_ERR_ARG, m);
    if (const char *m = check_list(fr.clist, 0)) return qr_fail(QR_ERR_ARG, m);
    for (int i = 0; i < n_srf; i++)
    {
        const qr_surface &s = v.srf[i];
        if (s.trnode != QR_NULL && !ok_srf(s.trnode)) return qr_fail(QR_ERR_ARG, "trnode out of range");
        if (s.has_trm != 0 && s.trnode == QR_NULL && s.srf_t[3] >= 0 && s.srf_t[3] < QR_TAG_SURFACE_MAX)
            return qr_fail(QR_ERR_ARG, "transformed surface without trnode");
        const bool real = s.srf_t[3] >= 0 && s.srf_t[3] < QR_TAG_SURFACE_MAX;
        for (int k = 0; k < 3; k++)
            if (((s.axes >> (2 * k)) & 3) > 2) return qr_fail(QR_ERR_ARG, "bad axis map");
        if (!real) continue;
        for (int k = 0; k < 2; k++)
            if (s.mat[k] < 0 || s.mat[k] >= n_mat) return qr_fail(QR_ERR_ARG, "material index out of range");
        if (!ok_elm(s.clip) || !ok_elm(s.lst[0]) || !ok_elm(s.lst[1]) || !ok_elm(s.lst[2]) || !ok_elm(s.lst[3]))
            return qr_fail(QR_ERR_ARG, "surface list head out of range");
        if (const char *m = check_list(s.clip, 1)) return qr_fail(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[0], 2)) return qr_fail(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[2], 2)) return qr_fail(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[1], 0)) return qr_fail(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[3], 0)) return qr_fail(QR_ERR_ARG, m);
        for (int side = 0; side < 2; side++)
            for (int e = s.lst[side * 2]; e != QR_NULL; e = v.elm[e].next)
                if (const char *m = check_list(v.elm[e].data, 0)) return qr_fail(QR_ERR_ARG, m);
    }

    /* ---- 1. build every device array on the host ---------------------------------------- */

    /* surfaces: repack qr_surface (256 B, snapshot layout) into DSurf (128 B, hot part first)
     * for the list walk and DShade for shading */
    std::vector<DSurf> dsurf(n_srf + 1);
    std::vector<DShade> dshade(n_srf + 1);
    memset(dsurf.data(), 0, dsurf.size() * sizeof(DSurf));
    memset(dshade.data(), 0, dshade.size() * sizeof(DShade));
    for (int i = 0; i < n_srf; i++)
    {
        const qr_surface &q = v.srf[i];
        const bool real = q.srf_t[3] >= 0 && q.srf_t[3] < QR_TAG_SURFACE_MAX;
        if (real && q.smask != QR_SMASK) return qr_fail(QR_ERR_ARG, "surface smask is not the fp32 sign bit");
        if (real && ((q.shift != 0) != (q.has_trm != 0)))
            return qr_fail(QR_ERR_UNSUP, "surface with trnode shift but no transform flags (or the reverse)");
        if ((q.conic & ~3) || (q.has_trm & ~3) || (q.srf_t[0] & ~3) || (q.srf_t[1] & ~3) || (q.srf_t[2] & ~3))
            return qr_fail(QR_ERR_ARG, "surface tag fields out of range");
        DSurf &d = dsurf[i];
        DShade &h = dshade[i];
        for (int k = 0; k < 3; k++)
        {
            d.pos[k] = q.pos[k]; d.scj[k] = q.scj[k];
            /* an axis without clipping gets an infinite bound: the kernel compares unconditionally */
            d.min[k] = (q.minmax_t & (1u << k)) ? q.min[k] : -__builtin_inff();
            d.max[k] = (q.minmax_t & (1u << (3 + k))) ? q.max[k] : __builtin_inff();
            d.tci[k] = q.tci[k]; d.tcj[k] = q.tcj[k]; d.tck[k] = q.tck[k];
        }
        for (int k = 0; k < 4; k++) d.sci[k] = q.sci[k];
        d.clip = q.clip; d.d_eps = q.d_eps; d.t_eps = q.t_eps;
        d.trnode = q.trnode;
        d.props0 = q.props[0]; d.props1 = q.props[1];
        h.mat[0] = q.mat[0] >= 0 ? q.mat[0] : 0; h.mat[1] = q.mat[1] >= 0 ? q.mat[1] : 0;
        uint32_t f = 0;
        f |= q.minmax_t & 63u;
        f |= ((uint32_t)q.conic & 3u) << 6;
        f |= ((uint32_t)q.has_trm & 3u) << 8;
        f |= (q.shift ? 1u : 0u) << 10;
        f |= ((q.axes >> 0) & 3u) << 11; f |= ((q.axes >> 2) & 3u) << 13; f |= ((q.axes >> 4) & 3u) << 15;
        f |= ((q.axes >> 8) & 7u) << 17;
        f |= (real ? ((uint32_t)q.srf_t[0] & 3u) : 0u) << 20;
        f |= ((uint32_t)q.srf_t[1] & 3u) << 22;
        f |= ((uint32_t)q.srf_t[2] & 3u) << 24;
        f |= (q.srf_t[3] < 0 ? 1u : 0u) << 26;
        f |= (q.c_def != 0 ? 1u : 0u) << 28;
        d.flags = f;
    }

    /* bounding spheres + cull flag (bit 2 of a surface-list cell's kind) */
    std::vector<BSphere> bsph(n_srf + 1);
    memset(bsph.data(), 0, bsph.size() * sizeof(BSphere));
    for (int i = 0; i < n_srf; i++) bsph[i] = bound_sphere(v, i);
    if (getenv("QR_VERBOSE"))
    {
        int nreal = 0, nfin = 0;
        for (int i = 0; i < n_srf; i++)
        {
            const qr_surface &q = v.srf[i];
            if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
            nreal++; if (bsph[i].r < 1e30f) nfin++;
        }
        fprintf(stderr, "bounding spheres: %d of %d real surfaces bounded\n", nfin, nreal);
    }
    std::vector<qr_elem> cells(n_elm + 1);
    memset(cells.data(), 0, cells.size() * sizeof(qr_elem));
    memcpy(cells.data(), v.elm, (size_t)n_elm * sizeof(qr_elem));
    {
        std::vector<uint8_t> seen(n_elm + 1, 0);
        const char *cm = getenv("QR_CULL");                 /* 0 off, 1 planes, 2 planes + open quadrics, 3 all */
        const int cull_mode = getenv("QR_NOCULL") ? 0 : (cm ? atoi(cm) : 3);
        auto mark_list = [&](int head) {
            for (int e = head; e != QR_NULL && !seen[e]; e = v.elm[e].next)
            {
                seen[e] = 1;
                const int si = v.elm[e].simd;
                const qr_surface &q = v.srf[si];
                const bool real = q.srf_t[3] >= 0 && q.srf_t[3] < QR_TAG_SURFACE_MAX;
                /* the solver already rejects a ray that misses a closed quadric as cheaply as the sphere test
                 * does; the test pays for planes and open quadrics, whose hits die only in the clippers */
                const bool open_shape = q.srf_t[0] == 1 || !(q.sci[0] > 0.0f && q.sci[1] > 0.0f && q.sci[2] > 0.0f);
                const bool want = cull_mode >= 3 || (cull_mode == 2 && open_shape) || (cull_mode == 1 && q.srf_t[0] == 1);
                if (real && (v.elm[e].kind & 3) == 0 && bsph[si].r < 1e30f && want) cells[e].kind |= 4;
            }
        };
        for (uint32_t i = 0; i < v.hdr->n_tiles; i++) mark_list(v.tiles[i]);
        mark_list(fr.clist);
        for (int i = 0; i < n_srf; i++)
        {
            const qr_surface &q = v.srf[i];
            if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
            mark_list(q.lst[1]); mark_list(q.lst[3]);
            for (int side = 0; side < 2; side++)
                for (int e = q.lst[side * 2]; e != QR_NULL; e = v.elm[e].next) mark_list(v.elm[e].data);
        }
    }
    for (int i = 0; i < n_srf; i++) for (int k = 0; k < 4; k++) dshade[i].lst[k] = v.srf[i].lst[k];

    /* wave schedule: one entry per wave footprint (8x8 / 8x4 / 4x4 pixels) */
    const int fw = fr.fsaa == 2 ? 4 : 8, fh = fr.fsaa == 0 ? 8 : 4;
    const int nbx = (fr.frm_w + fw - 1) / fw, nby = (fr.frm_h + fh - 1) / fh;
    if (nbx > 0x3FFF || nby > 0x3FFF) return qr_fail(QR_ERR_ARG, "frame too large");
    std::vector<uint32_t> order;
    {
        /* heavy = the footprint's tile list holds a reflective or non-opaque surface */
        std::vector<uint8_t> tile_heavy((size_t)fr.tls_row * fr.tls_col, 0);
        for (size_t t = 0; t < tile_heavy.size(); t++)
            for (int e = v.tiles[t]; e != QR_NULL; e = v.elm[e].next)
            {
                const qr_surface &q = v.srf[v.elm[e].simd];
                if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
                for (int k = 0; k < 2; k++)
                {
                    if (q.props[k] & QR_PROP_REFLECT) tile_heavy[t] |= 1;
                    if (!(q.props[k] & QR_PROP_OPAQUE)) tile_heavy[t] |= 2;
                }
            }
        std::vector<uint32_t> heavy, light;
        /* enumerate footprints tile by tile (32x8 pixel groups) to keep neighbours together */
        const int gx = 32 / fw, gy = 8 / fh;
        for (int ty = 0; ty * gy < nby; ty++)
            for (int tx = 0; tx * gx < nbx; tx++)
                for (int j = 0; j < gy; j++)
                    for (int i = 0; i < gx; i++)
                    {
                        const int bx = tx * gx + i, by = ty * gy + j;
                        if (bx >= nbx || by >= nby) continue;
                        const int tlx = (bx * fw) / fr.tile_w, tly = (by * fh) / fr.tile_h;
                        const int hv = (tlx < fr.tls_row && tly < fr.tls_col) ? tile_heavy[(size_t)tly * fr.tls_row + tlx] : 0;
                        const uint32_t ent = (uint32_t)bx | ((uint32_t)by << 14);
                        (hv ? heavy : light).push_back(ent);
                    }
        order = heavy;
        order.insert(order.end(), light.begin(), light.end());
    }

    /* ---- 2. one device allocation; arrays padded by one zero record so that masked-off
     *         lanes may read index 0 of an empty array ------------------------------------ */
    size_t o_srf = 0;
    size_t o_shd = pad16(o_srf + dsurf.size() * sizeof(DSurf));
    size_t o_mat = pad16(o_shd + dshade.size() * sizeof(DShade));
    size_t o_lgt = pad16(o_mat + (size_t)(n_mat + 1) * sizeof(qr_material));
    size_t o_elm = pad16(o_lgt + (size_t)(n_lgt + 1) * sizeof(qr_light));
    size_t o_til = pad16(o_elm + cells.size() * sizeof(DCell));
    size_t o_tex = pad16(o_til + (size_t)(v.hdr->n_tiles + 1) * 4);
    size_t o_ord = pad16(o_tex + (size_t)(n_tex + 1) * 4);
    size_t o_frm = pad16(o_ord + order.size() * 4 + 16);
    size_t o_bs = pad16(o_frm + sizeof(qr_frame));
    size_t total = pad16(o_bs + bsph.size() * sizeof(BSphere));

    std::vector<uint8_t> host(total, 0);
    memcpy(host.data() + o_srf, dsurf.data(), dsurf.size() * sizeof(DSurf));
    memcpy(host.data() + o_shd, dshade.data(), dshade.size() * sizeof(DShade));
    memcpy(host.data() + o_mat, v.mat, (size_t)n_mat * sizeof(qr_material));
    memcpy(host.data() + o_lgt, v.lgt, (size_t)n_lgt * sizeof(qr_light));
    {
        std::vector<DCell> dc(cells.size());
        for (size_t i = 0; i < cells.size(); i++)
        {
            DCell &c = dc[i];
            c.simd = cells[i].simd; c.data = cells[i].data; c.next = cells[i].next; c.kind = cells[i].kind;
            c.cx = c.cy = c.cz = 0.0f; c.r = __builtin_inff();
            if ((c.kind & 4) && c.simd >= 0 && c.simd < n_srf)
            {
                const BSphere &b = bsph[c.simd];
                c.cx = b.c[0]; c.cy = b.c[1]; c.cz = b.c[2]; c.r = b.r;
            }
        }
        memcpy(host.data() + o_elm, dc.data(), dc.size() * sizeof(DCell));
    }
    memcpy(host.data() + o_til, v.tiles, (size_t)v.hdr->n_tiles * 4);
    memcpy(host.data() + o_tex, v.texels, (size_t)n_tex * 4);
    memcpy(host.data() + o_ord, order.data(), order.size() * 4);
    memcpy(host.data() + o_frm, &fr, sizeof(qr_frame));
    memcpy(host.data() + o_bs, bsph.data(), bsph.size() * sizeof(BSphere));

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return qr_fail(QR_ERR_DEVICE, "no HIP device available (the gfx950 backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return qr_fail(QR_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    qr_device_scene *s = new qr_device_scene();
    memset(s, 0, sizeof(*s));
    s->device = device;
    s->hdr = *v.hdr;
    hipError_t e = hipMalloc(&s->d_blob, total);
    if (e != hipSuccess) { delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(s->d_blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
#ifdef QR_WAVETIME
    e = hipMalloc((void **)&s->d_counters, (32 + 4 * order.size()) * sizeof(unsigned long long));
#else
    e = hipMalloc((void **)&s->d_counters, 32 * sizeof(unsigned long long));
#endif
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    (void)hipEventCreate(&s->ev0);
    (void)hipEventCreate(&s->ev1);
    s->blob_bytes = total;

    uint8_t *d = (uint8_t *)s->d_blob;
    s->sc.srf = (const DSurf *)(d + o_srf);
    s->sc.shd = (const DShade *)(d + o_shd);
    s->sc.mat = (const qr_material *)(d + o_mat);
    s->sc.lgt = (const qr_light *)(d + o_lgt);
    s->sc.elm = (const DCell *)(d + o_elm);
    s->sc.tiles = (const int32_t *)(d + o_til);
    s->sc.texels = (const uint32_t *)(d + o_tex);
    s->sc.bsph = (const void *)(d + o_bs);
    s->sc.order = (const uint32_t *)(d + o_ord);
    s->sc.n_blocks = (int32_t)order.size();
    s->sc.stats = s->d_counters + 4;
    s->sc.frp = (const qr_frame *)(d + o_frm);
    s->fr = fr;
    s->sc.depth = fr.depth > QR_MAX_DEPTH ? QR_MAX_DEPTH : fr.depth;
    s->sc.row_begin = 0; s->sc.row_end = fr.frm_h;
    s->sc.index = fr.index; s->sc.thnum = fr.thnum > 0 ? fr.thnum : 1;
    s->sc.group_first = 0; s->sc.group_stride = 1;
    s->sc.n_groups = (fr.frm_h + 7) / 8;
    s->sc.dbg = getenv("QR_DBG") ? atoi(getenv("QR_DBG")) : 0;
    *out = s;
    return QR_OK;
}

extern "C" int qr_scene_destroy(qr_device_scene *s)
{
    if (s == nullptr) return QR_OK;
    (void)hipSetDevice(s->device);
    (void)hipEventDestroy(s->ev0);
    (void)hipEventDestroy(s->ev1);
    (void)hipFree(s->d_counters);
    (void)hipFree(s->d_blob);
    delete s;
    return QR_OK;
}

extern "C" int qr_scene_get_info(const qr_device_scene *s, qr_scene_info *info)
{
    if (s == nullptr || info == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    memset(info, 0, sizeof(*info));
    info->frm_w = s->fr.frm_w; info->frm_h = s->fr.frm_h;
    info->fsaa = s->fr.fsaa; info->depth = s->sc.depth;
    info->n_srf = (int32_t)s->hdr.n_srf; info->n_mat = (int32_t)s->hdr.n_mat; info->n_lgt = (int32_t)s->hdr.n_lgt;
    info->n_elm = (int32_t)s->hdr.n_elm; info->n_tiles = (int32_t)s->hdr.n_tiles; info->n_texels = (int32_t)s->hdr.n_texels;
    info->tile_w = s->fr.tile_w; info->tile_h = s->fr.tile_h;
    info->device_bytes = s->blob_bytes;
    return QR_OK;
}

extern "C" int qr_scene_set_depth(qr_device_scene *s, int depth)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    if (depth < 0 || depth > QR_MAX_DEPTH) return qr_fail(QR_ERR_ARG, "depth must be 0..10 (RT_STACK_DEPTH)");
    s->sc.depth = depth;
    return QR_OK;
}

extern "C" int qr_scene_set_rows(qr_device_scene *s, int row_begin, int row_end, int index, int thnum)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    const int h = s->fr.frm_h;
    if (row_begin < 0 || row_end > h || row_begin > row_end) return qr_fail(QR_ERR_ARG, "bad row range");
    if (thnum <= 0 || index < 0 || index >= thnum) return qr_fail(QR_ERR_ARG, "bad index/thnum");
    s->sc.row_begin = row_begin; s->sc.row_end = row_end;
    s->sc.index = index; s->sc.thnum = thnum;
    s->sc.group_first = row_begin / 8;
    s->sc.group_stride = 1;
    s->sc.n_groups = row_end > row_begin ? (row_end - 1) / 8 - row_begin / 8 + 1 : 0;
    return QR_OK;
}

extern "C" int qr_scene_set_tile_rows(qr_device_scene *s, int first, int stride)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    const int total = (s->fr.frm_h + 7) / 8;
    if (stride <= 0 || first < 0) return qr_fail(QR_ERR_ARG, "bad tile-row selection");
    s->sc.row_begin = 0; s->sc.row_end = s->fr.frm_h;
    s->sc.index = 0; s->sc.thnum = 1;
    s->sc.group_first = first; s->sc.group_stride = stride;
    s->sc.n_groups = first < total ? (total - first + stride - 1) / stride : 0;
    return QR_OK;
}

template <bool COUNT>
static hipError_t launch(qr_device_scene *s, void *frame_dev, int32_t *ids_dev, hipStream_t st)
{
    const int fsaa = s->fr.fsaa;
    const int bw = fsaa == 0 ? 32 : fsaa == 1 ? 16 : 8;
    (void)bw;
    dim3 grid((s->sc.n_blocks + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64), 1, 1);
    if (grid.x == 0 || s->sc.n_groups == 0) return hipSuccess;
    /* register budget variant (waves per SIMD); QR_WAVES is a tuning knob for experiments */
    static const int waves = []() { const char *e = getenv("QR_WAVES"); int w = e ? atoi(e) : QR_MIN_WAVES_PER_SIMD;
                                    return (w == 2 || w == 3 || w == 4) ? w : QR_MIN_WAVES_PER_SIMD; }();
    uint32_t *f = (uint32_t *)frame_dev;
    if (waves == 4)      hipLaunchKernelGGL((qr_render_kernel<COUNT, 4>), grid, dim3(QR_BLOCK), 0, st, s->sc, f, ids_dev, s->d_counters);
    else if (waves == 3) hipLaunchKernelGGL((qr_render_kernel<COUNT, 3>), grid, dim3(QR_BLOCK), 0, st, s->sc, f, ids_dev, s->d_counters);
    else                 hipLaunchKernelGGL((qr_render_kernel<COUNT, 2>), grid, dim3(QR_BLOCK), 0, st, s->sc, f, ids_dev, s->d_counters);
    return hipGetLastError();
}

extern "C" int qr_render_async(qr_device_scene *s, void *frame_dev, void *stream)
{
    if (s == nullptr || frame_dev == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch<false>(s, frame_dev, nullptr, (hipStream_t)stream));
    return QR_OK;
}

extern "C" int qr_render_ids_async(qr_device_scene *s, void *frame_dev, void *ids_dev, void *stream)
{
    if (s == nullptr || frame_dev == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch<false>(s, frame_dev, (int32_t *)ids_dev, (hipStream_t)stream));
    return QR_OK;
}

extern "C" int qr_render_count(qr_device_scene *s, void *frame_dev, void *stream, qr_ray_counts *counts)
{
    if (s == nullptr || frame_dev == nullptr || counts == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, 32 * sizeof(unsigned long long), st));
    HIP_TRY(launch<true>(s, frame_dev, nullptr, st));
    unsigned long long h[4];
    HIP_TRY(hipMemcpyAsync(h, s->d_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    counts->primary = h[0]; counts->shadow = h[1]; counts->reflect = h[2]; counts->refract = h[3];
#ifdef QR_STATS2
    {
        unsigned long long st[12];
        HIP_TRY(hipMemcpy(st, s->d_counters + 4, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "QR_STATS2 clip calls %llu: cycles per call: depth/hit/conic/minmax %.0f, custom clippers %.0f\n", st[11], st[9] / (st[11] + 1e-9), st[10] / (st[11] + 1e-9));
        const double ni = st[3] + 1e-9, nf = st[4] + 1e-9;
        fprintf(stderr, "QR_STATS2 shadow walks: iterations %llu (full %llu): cycles/iteration: cell load %.0f, cull %.0f; per full element %.0f = hot load %.0f + diff/transform %.0f + solver %.0f + candidates/clip %.0f\n",
                st[3], st[4], st[0] / ni, st[1] / ni, st[2] / nf, st[5] / nf, st[6] / nf, st[7] / nf, st[8] / nf);
    }
#endif
#ifdef QR_STATS
    {
        unsigned long long st[16];
        HIP_TRY(hipMemcpy(st, s->d_counters + 4, sizeof(st), hipMemcpyDeviceToHost));
        const char *nm[3] = { "shadow", "primary", "secondary" };
        for (int k = 0; k < 3; k++)
            fprintf(stderr, "QR_STATS %s: walks %llu elem-iterations %llu (%.1f per walk, %.0f%% culled) active lanes per iteration %.1f\n", nm[k],
                    st[3 * k], st[3 * k + 1], st[3 * k] ? (double)st[3 * k + 1] / st[3 * k] : 0.0,
                    st[3 * k + 1] ? 100.0 * st[12 + k] / st[3 * k + 1] : 0.0,
                    st[3 * k + 1] ? (double)st[3 * k + 2] / st[3 * k + 1] : 0.0);
        fprintf(stderr, "QR_STATS wave-cycles (s_memtime): primary/secondary traverse %llu, shade incl. shadow walks %llu, rest %llu\n", st[9], st[10], st[11]);
    }
#endif
    return QR_OK;
}

extern "C" int qr_render_timed(qr_device_scene *s, void *frame_dev, void *stream,
                               int iters, float *avg_ms, float *min_ms)
{
    if (s == nullptr || frame_dev == nullptr || iters <= 0) return qr_fail(QR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(s->device));
    double sum = 0.0; float mn = 1e30f;
    for (int i = 0; i < iters; i++)
    {
        HIP_TRY(hipEventRecord(s->ev0, st));
        HIP_TRY(launch<false>(s, frame_dev, nullptr, st));
        HIP_TRY(hipEventRecord(s->ev1, st));
        HIP_TRY(hipEventSynchronize(s->ev1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        sum += ms; if (ms < mn) mn = ms;
    }
    if (avg_ms) *avg_ms = (float)(sum / iters);
    if (min_ms) *min_ms = mn;
#ifdef QR_WAVETIME
    if (const char *path = getenv("QR_WAVETIME_OUT"))
    {
        /* per wave of the last launch: {start, first traverse done, end} in 100 MHz ticks, {hw_id | xcc << 32 | walks << 40} */
        std::vector<unsigned long long> w((size_t)s->sc.n_blocks * 4);
        HIP_TRY(hipMemcpy(w.data(), s->d_counters + 32, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        FILE *f = fopen(path, "wb");
        if (f) { fwrite(w.data(), sizeof(unsigned long long), w.size(), f); fclose(f); }
    }
#endif
    return QR_OK;
}

extern "C" int qr_render_host(qr_device_scene *s, uint32_t *frame_host, int row_pixels)
{
    if (s == nullptr || frame_host == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    const int w = s->fr.frm_w, h = s->fr.frm_h;
    HIP_TRY(hipSetDevice(s->device));
    void *d_frame = nullptr;
    HIP_TRY(hipMalloc(&d_frame, (size_t)w * h * 4));
    int rc = QR_OK;
    hipError_t e = launch<false>(s, d_frame, nullptr, nullptr);
    std::vector<uint32_t> tmp((size_t)w * h);
    if (e == hipSuccess) e = hipMemcpy(tmp.data(), d_frame, (size_t)w * h * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_frame);
    if (e != hipSuccess) return qr_fail(QR_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
    /* copy only the rows this call owns, honouring a negative stride (bottom-up
     * frames, engine.cpp:2814-2850) */
    for (int y = s->sc.row_begin; y < s->sc.row_end; y++)
    {
        if ((y / 8 - s->sc.group_first) % s->sc.group_stride != 0 || y / 8 < s->sc.group_first) continue;
        if (s->sc.thnum > 1 && (y % s->sc.thnum) != s->sc.index) continue;
        memcpy(frame_host + (ptrdiff_t)y * row_pixels, tmp.data() + (size_t)y * w, (size_t)w * 4);
    }
    return rc;
}

/*
 * The reference entry point.  One call = flatten + upload + launch + copy back.
 * The scene is re-flattened every call because the engine rebuilds its lists
 * and animates objects every frame (engine.cpp:2976-3332); all of that is a few
 * hundred KB.
 */
extern "C" int qr_render0(const void *s_inf, const qr_abi_desc *abi)
{
    std::vector<uint8_t> blob;
    std::string err;
    int rc = qr_flatten_impl(s_inf, abi, blob, err);
    if (rc != QR_OK) return qr_fail(rc, err);

    /* frame pointer and stride: inf_FRAME / inf_FRM_ROW, tracer.h:186-190 */
    const uint8_t *inf = (const uint8_t *)s_inf;
    const size_t ps = abi->pointer_bits / 8;
    const size_t ib = (size_t)abi->quads * 0x100;
    uint64_t p_frame = 0; int64_t row = 0;
    if (ps == 8) { memcpy(&p_frame, inf + ib + 11 * ps, 8); memcpy(&row, inf + ib + 10 * ps, 8); }
    else { uint32_t a; int32_t b; memcpy(&a, inf + ib + 11 * ps, 4); memcpy(&b, inf + ib + 10 * ps, 4); p_frame = a; row = b; }
    if (p_frame == 0) return qr_fail(QR_ERR_ARG, "s_inf->frame is NULL");

    int dev = 0;
    if (const char *env = getenv("QR_DEVICE")) dev = atoi(env);
    qr_device_scene *scn = nullptr;
    rc = qr_scene_upload(blob.data(), blob.size(), dev, &scn);
    if (rc != QR_OK) return rc;
    rc = qr_render_host(scn, (uint32_t *)(uintptr_t)p_frame, (int)row);
    qr_scene_destroy(scn);
    return rc;
}
