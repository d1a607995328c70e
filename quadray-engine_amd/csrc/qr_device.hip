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
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <vector>
#include <mutex>
#include <map>
#include <array>
#include <algorithm>
#include <chrono>
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
    int device = 0;
    void *d_blob = nullptr;     /* one allocation holding every array      */
    uint64_t blob_bytes = 0;
    DevScene sc = {};           /* device pointers + launch parameters      */
    qr_frame fr = {};           /* host copy of the frame parameters        */
    qr_header hdr = {};
    unsigned long long *d_counters = nullptr;
    size_t n_cells = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    /* the whole-frame wave schedule (host copy) and the schedules of the row selections rendered so far:
     * a launch restricted by qr_scene_set_rows / _set_tile_rows only starts the waves that own pixels */
    bool divergent = false;     /* launch the per-lane (divergent) walk variant, see walk_div */
    std::vector<uint32_t> h_order;
    struct SubSched { uint32_t *d_order; int32_t n; };
    std::map<std::array<int32_t, 6>, SubSched> sub;
};

static void multi_forget(const qr_device_scene *s);

extern "C" const char *qr_kernel_name(void) { return "qr_render_kernel"; }

extern "C" int qr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static size_t pad16(size_t x) { return (x + 15) & ~(size_t)15; }

/* Tried and dropped for the drop-in path: page-locking the engine's frame in place for a direct DMA
 * copy-back saves 0.35 ms per 1080p frame on demo1 but makes demo2 ten times slower (host accesses to the
 * engine's heap next to the frame and later copies slow down once the range is registered); recycling the
 * scene's device allocation between calls showed the same effect. */

#include "qr_bounds.hpp"
#include "qr_binning.hpp"

extern "C" int qr_scene_upload(const void *blob, uint64_t size, int device, qr_device_scene **out)
{
    return qr_scene_upload_ex(blob, size, device, 0u, out);
}

extern "C" int qr_scene_upload_ex(const void *blob, uint64_t size, int device, uint32_t flags, qr_device_scene **out)
{
    if (blob == nullptr || out == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    const bool ph_verbose = getenv("QR_VERBOSE") && atoi(getenv("QR_VERBOSE")) >= 2;
    double ph_t = now_ms();
    auto phase = [&](const char *name) { if (ph_verbose) { const double t = now_ms(); fprintf(stderr, "upload phase %-12s %.3f ms\n", name, t - ph_t); ph_t = t; } };
    if (getenv("QR_REBIN") && atoi(getenv("QR_REBIN")) != 0) flags |= QR_UPLOAD_REBIN_TILES;
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
    std::vector<uint8_t> checked((size_t)n_elm + 1, 0);     /* bit k: head already validated as a list of kind k */
    /* are the bounding-volume arrays of every surface list properly nested (each array ends inside the
     * array that contains its head)?  The engine builds them that way; the kernel's packet jump over an
     * array relies on it, so it is checked here and the jump narrowed when it does not hold. */
    bool nested = true;
    std::vector<int> pos_stamp((size_t)n_elm + 1, -1), pos_idx((size_t)n_elm + 1, 0);
    int stamp = 0;
    auto check_list = [&](int head, int kind) -> const char * {
        /* kind 0 surfaces, 1 clippers, 2 lights */
        if (head == QR_NULL) return nullptr;
        if (checked[head] & (1u << kind)) return nullptr;     /* shared lists (one global list per scene) are walked once */
        checked[head] |= (uint8_t)(1u << kind);
        int cnt = 0;
        for (int e = head; e != QR_NULL; e = v.elm[e].next)
        {
            if (++cnt > n_elm) return "cyclic list";
            const qr_elem &el = v.elm[e];
            if (kind == 0) { pos_stamp[e] = stamp; pos_idx[e] = cnt; }
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
        if (kind == 0)
        {
            std::vector<int> open_end;          /* list positions at which the open arrays end */
            int p = 0;
            for (int e = head; e != QR_NULL; e = v.elm[e].next)
            {
                p++;
                while (!open_end.empty() && open_end.back() < p) open_end.pop_back();
                const qr_elem &el = v.elm[e];
                if ((el.kind & 3) == 1)
                {
                    if (el.data == QR_NULL || pos_stamp[el.data] != stamp || pos_idx[el.data] < p) { nested = false; break; }
                    if (!open_end.empty() && pos_idx[el.data] > open_end.back()) { nested = false; break; }
                    open_end.push_back(pos_idx[el.data]);
                }
            }
            stamp++;
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

    /* working copies: the tile lists (and with them the cell array and the tile geometry of the
     * frame record) are replaced when the binning pass runs */
    std::vector<qr_elem> E(v.elm, v.elm + n_elm);
    std::vector<int32_t> T(v.tiles, v.tiles + v.hdr->n_tiles);
    qr_frame frm = *v.frame;

    phase("validate");
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

    phase("surfaces");
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
    if (flags & QR_UPLOAD_REBIN_TILES)
    {
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
            return qr_fail(QR_ERR_DEVICE, "no HIP device available (the gfx950 backend has no CPU fallback)");
        if (device < 0 || device >= nd) return qr_fail(QR_ERR_ARG, "device ordinal out of range");
        HIP_TRY(hipSetDevice(device));
        rc = rebin_tiles(v, bsph, frm, E, T);
        if (rc != QR_OK) return rc;
    }
    std::vector<qr_elem> cells(E.size() + 1);
    memset(cells.data(), 0, cells.size() * sizeof(qr_elem));
    memcpy(cells.data(), E.data(), E.size() * sizeof(qr_elem));
    {
        std::vector<uint8_t> seen(E.size() + 1, 0);
        const char *cm = getenv("QR_CULL");                 /* 0 off, 1 planes, 2 planes + open quadrics, 3 all */
        const int cull_mode = getenv("QR_NOCULL") ? 0 : (cm ? atoi(cm) : 3);
        auto mark_list = [&](int head) {
            for (int e = head; e != QR_NULL && !seen[e]; e = E[e].next)
            {
                seen[e] = 1;
                const int si = E[e].simd;
                const qr_surface &q = v.srf[si];
                const bool real = q.srf_t[3] >= 0 && q.srf_t[3] < QR_TAG_SURFACE_MAX;
                /* the solver already rejects a ray that misses a closed quadric as cheaply as the sphere test
                 * does; the test pays for planes and open quadrics, whose hits die only in the clippers */
                const bool open_shape = q.srf_t[0] == 1 || !(q.sci[0] > 0.0f && q.sci[1] > 0.0f && q.sci[2] > 0.0f);
                const bool want = cull_mode >= 3 || (cull_mode == 2 && open_shape) || (cull_mode == 1 && q.srf_t[0] == 1);
                if (real && (E[e].kind & 3) == 0 && bsph[si].r < 1e30f && want) cells[e].kind |= 4;
            }
        };
        for (uint32_t i = 0; i < (uint32_t)T.size(); i++) mark_list(T[i]);
        mark_list(frm.clist);
        for (int i = 0; i < n_srf; i++)
        {
            const qr_surface &q = v.srf[i];
            if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
            mark_list(q.lst[1]); mark_list(q.lst[3]);
            for (int side = 0; side < 2; side++)
                for (int e = q.lst[side * 2]; e != QR_NULL; e = E[e].next) mark_list(E[e].data);
        }
    }
    for (int i = 0; i < n_srf; i++) for (int k = 0; k < 4; k++) dshade[i].lst[k] = v.srf[i].lst[k];

    phase("bounds+cells");
    /* wave schedule: one entry per wave footprint (8x8 / 8x4 / 4x4 pixels) */
    const int fw = frm.fsaa == 2 ? 4 : 8, fh = frm.fsaa == 0 ? 8 : 4;
    const int nbx = (frm.frm_w + fw - 1) / fw, nby = (frm.frm_h + fh - 1) / fh;
    if (nbx > 0x3FFF || nby > 0x3FFF) return qr_fail(QR_ERR_ARG, "frame too large");
    std::vector<uint32_t> order;
    {
        /* heavy = the footprint's tile list holds a reflective or non-opaque surface */
        std::vector<uint8_t> tile_heavy((size_t)frm.tls_row * frm.tls_col, 0);
        for (size_t t = 0; t < tile_heavy.size(); t++)
            for (int e = T[t]; e != QR_NULL; e = E[e].next)
            {
                const qr_surface &q = v.srf[E[e].simd];
                if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
                for (int k = 0; k < 2; k++)
                {
                    if (q.props[k] & QR_PROP_REFLECT) tile_heavy[t] |= 1;
                    if (!(q.props[k] & QR_PROP_OPAQUE)) tile_heavy[t] |= 2;
                }
            }
        /* enumerate footprints tile by tile (32x8 pixel groups) to keep neighbours together; entries are
         * {footprint | heaviness, tile-list head}: the head when all pixels of the footprint lie in one tile
         * (always, for the engine's 32x8 tiles), so that the wave needs no per-lane tile lookup.  This runs
         * every frame in the drop-in path: no divisions in the common case, no reallocation. */
        const int gx = 32 / fw, gy = 8 / fh;
        const bool nest = frm.tile_w == 32 && frm.tile_h == 8;      /* group (tx, ty) IS tile (tx, ty) */
        std::vector<uint32_t> hv_ent, lt_ent;
        hv_ent.reserve((size_t)nbx * nby / 4 + 16); lt_ent.reserve((size_t)nbx * nby * 2 + 16);
        for (int ty = 0; ty * gy < nby; ty++)
            for (int tx = 0; tx * gx < nbx; tx++)
            {
                int g_hv = 0; int32_t g_head = QR_PER_LANE_TILE;
                if (nest && tx < frm.tls_row && ty < frm.tls_col)
                {
                    const size_t t = (size_t)ty * frm.tls_row + tx;
                    g_hv = tile_heavy[t]; g_head = T[t];
                }
                for (int j = 0; j < gy; j++)
                    for (int i = 0; i < gx; i++)
                    {
                        const int bx = tx * gx + i, by = ty * gy + j;
                        if (bx >= nbx || by >= nby) continue;
                        int hv = g_hv; int32_t head = g_head;
                        if (!nest)
                        {
                            const int x0 = bx * fw, y0 = by * fh;
                            const int x1 = std::min(x0 + fw - 1, frm.frm_w - 1), y1 = std::min(y0 + fh - 1, frm.frm_h - 1);
                            const int tlx = x0 / frm.tile_w, tly = y0 / frm.tile_h;
                            const bool in = tlx < frm.tls_row && tly < frm.tls_col;
                            hv = in ? tile_heavy[(size_t)tly * frm.tls_row + tlx] : 0;
                            head = QR_PER_LANE_TILE;
                            if (in && tlx == x1 / frm.tile_w && tly == y1 / frm.tile_h) head = T[(size_t)tly * frm.tls_row + tlx];
                        }
                        const uint32_t ent = (uint32_t)bx | ((uint32_t)by << 14) | ((uint32_t)(hv & 3) << 30);
                        std::vector<uint32_t> &dst = hv ? hv_ent : lt_ent;
                        dst.push_back(ent); dst.push_back((uint32_t)head);
                    }
            }
        order.swap(hv_ent);
        order.insert(order.end(), lt_ent.begin(), lt_ent.end());
    }
    const size_t n_sched = order.size() / 2;

    phase("schedule");
    /* ---- 2. one device allocation; arrays padded by one zero record so that masked-off
     *         lanes may read index 0 of an empty array ------------------------------------ */
    size_t o_srf = 0;
    size_t o_shd = pad16(o_srf + dsurf.size() * sizeof(DSurf));
    size_t o_mat = pad16(o_shd + dshade.size() * sizeof(DShade));
    size_t o_lgt = pad16(o_mat + (size_t)(n_mat + 1) * sizeof(qr_material));
    size_t o_elm = pad16(o_lgt + (size_t)(n_lgt + 1) * sizeof(qr_light));
    size_t o_til = pad16(o_elm + cells.size() * sizeof(DCell));
    size_t o_tex = pad16(o_til + (size_t)((uint32_t)T.size() + 1) * 4);
    size_t o_ord = pad16(o_tex + (size_t)(n_tex + 1) * 4);
    size_t o_frm = pad16(o_ord + order.size() * 4 + 16);
    size_t o_bs = pad16(o_frm + sizeof(qr_frame));
    size_t total = pad16(o_bs + bsph.size() * sizeof(BSphere));

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return qr_fail(QR_ERR_DEVICE, "no HIP device available (the gfx950 backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return qr_fail(QR_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    /* staging buffer in page-locked memory, kept per thread: a pageable host-to-device copy above ~1 MB
     * takes 13-23 ms on this stack (the runtime pins the source on the fly), a pinned one 0.1 ms */
    static thread_local struct Staging { uint8_t *p = nullptr; size_t cap = 0; } stage;
    if (stage.cap < total)
    {
        if (stage.p) (void)hipHostFree(stage.p);
        stage.p = nullptr; stage.cap = 0;
        const size_t cap = total + total / 2;
        HIP_TRY(hipHostMalloc((void **)&stage.p, cap, hipHostMallocDefault));
        stage.cap = cap;
    }
    struct HostView { uint8_t *p; uint8_t *data() const { return p; } } host = { stage.p };
    memset(host.p, 0, total);
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
    memcpy(host.data() + o_til, T.data(), T.size() * 4);
    memcpy(host.data() + o_tex, v.texels, (size_t)n_tex * 4);
    memcpy(host.data() + o_ord, order.data(), order.size() * 4);
    memcpy(host.data() + o_frm, &frm, sizeof(qr_frame));
    memcpy(host.data() + o_bs, bsph.data(), bsph.size() * sizeof(BSphere));

    qr_device_scene *s = new qr_device_scene();     /* value-initialised: plain members are zero */
    s->device = device;
    s->hdr = *v.hdr;
    phase("stage");
    const double tm0 = now_ms();
    hipError_t e = hipMalloc(&s->d_blob, total);
    if (e != hipSuccess) { delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    const double tm1 = now_ms();
    e = hipMemcpy(s->d_blob, host.data(), total, hipMemcpyHostToDevice);
    if (getenv("QR_VERBOSE")) fprintf(stderr, "upload: device bytes %zu, hipMalloc %.3f ms, hipMemcpy %.3f ms\n", total, tm1 - tm0, now_ms() - tm1);
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
#ifdef QR_WAVETIME
    e = hipMalloc((void **)&s->d_counters, (32 + QR_WT_SLOTS * n_sched) * sizeof(unsigned long long));
#else
    e = hipMalloc((void **)&s->d_counters, 32 * sizeof(unsigned long long));
#endif
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    s->blob_bytes = total;
    phase("device");

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
    s->sc.n_blocks = (int32_t)n_sched;
    s->h_order = order;
    s->sc.nested = nested ? 1 : 0;
    {
        /* Divergent variant (walk_div): opt-in with QR_DIV=1.  Bit-exact on every fixture, but an iteration
         * made of dependent vector loads costs several scalar ones: it shortens the deepest waves' chains
         * (demo2 1080p: 1.04 ms per isolated launch against 1.13) and loses on throughput everywhere measured,
         * including the 10 000-quadric scene once the per-lane clipper loop raised its register need to 154. */
        const char *dv = getenv("QR_DIV");
        s->divergent = dv != nullptr && atoi(dv) != 0;
    }
    s->sc.stats = s->d_counters + 4;
    s->sc.frp = (const qr_frame *)(d + o_frm);
    s->fr = frm;
    s->n_cells = E.size();
    s->sc.depth = frm.depth > QR_MAX_DEPTH ? QR_MAX_DEPTH : frm.depth;
    s->sc.row_begin = 0; s->sc.row_end = frm.frm_h;
    s->sc.index = frm.index; s->sc.thnum = frm.thnum > 0 ? frm.thnum : 1;
    s->sc.group_first = 0; s->sc.group_stride = 1;
    s->sc.n_groups = (frm.frm_h + 7) / 8;
    s->sc.dbg = getenv("QR_DBG") ? atoi(getenv("QR_DBG")) : 0;
    *out = s;
    return QR_OK;
}

extern "C" int qr_scene_destroy(qr_device_scene *s)
{
    if (s == nullptr) return QR_OK;
    (void)hipSetDevice(s->device);
    multi_forget(s);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (auto &kv : s->sub) (void)hipFree(kv.second.d_order);
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
    info->n_elm = (int32_t)s->n_cells; info->n_tiles = s->fr.tls_row * s->fr.tls_col; info->n_texels = (int32_t)s->hdr.n_texels;
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
    if (s->sc.n_groups == 0) return hipSuccess;
    DevScene sc = s->sc;
    const int H = s->fr.frm_h;
    const bool whole = sc.row_begin == 0 && sc.row_end == H && sc.group_first == 0 && sc.group_stride == 1 && sc.thnum <= 1;
    if (!whole)
    {
        const std::array<int32_t, 6> key = { sc.row_begin, sc.row_end, sc.group_first, sc.group_stride, sc.index, sc.thnum };
        auto it = s->sub.find(key);
        if (it == s->sub.end())
        {
            /* first launch with this selection: keep the schedule entries whose footprint holds a selected row */
            const int fh = fsaa == 0 ? 8 : 4;
            std::vector<uint32_t> keep;
            for (size_t i = 0; i + 1 < s->h_order.size(); i += 2)
            {
                const int y0 = (int)((s->h_order[i] >> 14) & 0x3FFFu) * fh;
                bool any = false;
                for (int y = y0; y < y0 + fh && y < H && !any; y++)
                {
                    const int g = y >> 3;
                    any = y >= sc.row_begin && y < sc.row_end && g >= sc.group_first && (g - sc.group_first) % sc.group_stride == 0
                       && (sc.thnum <= 1 || (y % sc.thnum) == sc.index);
                }
                if (any) { keep.push_back(s->h_order[i]); keep.push_back(s->h_order[i + 1]); }
            }
            if (s->sub.size() >= 256)
            {
                /* a selection is still in use by queued launches: wait before recycling the buffers */
                (void)hipDeviceSynchronize();
                for (auto &kv : s->sub) (void)hipFree(kv.second.d_order);
                s->sub.clear();
            }
            qr_device_scene::SubSched ss = { nullptr, (int32_t)(keep.size() / 2) };
            if (ss.n > 0)
            {
                hipError_t e = hipMalloc((void **)&ss.d_order, keep.size() * 4);
                if (e != hipSuccess) return e;
                e = hipMemcpy(ss.d_order, keep.data(), keep.size() * 4, hipMemcpyHostToDevice);
                if (e != hipSuccess) { (void)hipFree(ss.d_order); return e; }
            }
            it = s->sub.emplace(key, ss).first;
        }
        sc.order = it->second.d_order;
        sc.n_blocks = it->second.n;
    }
    dim3 grid((sc.n_blocks + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64), 1, 1);
    if (grid.x == 0) return hipSuccess;
    /* register budget variant (waves per SIMD); QR_WAVES is a tuning knob for experiments */
    static const int waves = []() { const char *e = getenv("QR_WAVES"); int w = e ? atoi(e) : QR_MIN_WAVES_PER_SIMD;
                                    return (w == 2 || w == 3 || w == 4) ? w : QR_MIN_WAVES_PER_SIMD; }();
    uint32_t *f = (uint32_t *)frame_dev;
    if (s->divergent)    hipLaunchKernelGGL((qr_render_kernel<COUNT, 3, true>), grid, dim3(QR_BLOCK), 0, st, sc, f, ids_dev, s->d_counters);
    else if (waves == 4) hipLaunchKernelGGL((qr_render_kernel<COUNT, 4>), grid, dim3(QR_BLOCK), 0, st, sc, f, ids_dev, s->d_counters);
    else if (waves == 3) hipLaunchKernelGGL((qr_render_kernel<COUNT, 3>), grid, dim3(QR_BLOCK), 0, st, sc, f, ids_dev, s->d_counters);
    else                 hipLaunchKernelGGL((qr_render_kernel<COUNT, 2>), grid, dim3(QR_BLOCK), 0, st, sc, f, ids_dev, s->d_counters);
    return hipGetLastError();
}

extern "C" int qr_render_async(qr_device_scene *s, void *frame_dev, void *stream)
{
    if (s == nullptr || frame_dev == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch<false>(s, frame_dev, nullptr, (hipStream_t)stream));
    return QR_OK;
}

/* combined schedules of multi-target launches, keyed by (scene, row range) per target */
struct MultiSched { uint32_t *d_order; DevScene *d_scenes; int32_t n; };
static std::map<std::vector<int64_t>, MultiSched> g_multi;
static std::mutex g_multi_lock;

static void multi_forget(const qr_device_scene *s)
{
    std::lock_guard<std::mutex> lk(g_multi_lock);
    for (auto it = g_multi.begin(); it != g_multi.end(); )
    {
        bool uses = false;
        for (size_t i = 0; i < it->first.size(); i += 3) if (it->first[i] == (int64_t)(intptr_t)s) uses = true;
        if (uses) { (void)hipFree(it->second.d_order); (void)hipFree(it->second.d_scenes); it = g_multi.erase(it); }
        else ++it;
    }
}

extern "C" int qr_render_multi_async(int n, qr_device_scene *const *scenes, void *const *frames_dev,
                                     const int *row_begin, const int *row_end, void *stream)
{
    if (n <= 0 || scenes == nullptr || frames_dev == nullptr || row_begin == nullptr || row_end == nullptr)
        return qr_fail(QR_ERR_ARG, "bad argument");
    bool direct = n <= QR_MAX_TARGETS;
    for (int i = 0; i < n; i++)
    {
        if (scenes[i] == nullptr || frames_dev[i] == nullptr) return qr_fail(QR_ERR_ARG, "null scene or frame");
        if (scenes[i]->device != scenes[0]->device) return qr_fail(QR_ERR_ARG, "scenes live on different devices");
        if (row_begin[i] < 0 || row_end[i] > scenes[i]->fr.frm_h || row_begin[i] > row_end[i]) return qr_fail(QR_ERR_ARG, "bad row range");
        if (scenes[i]->divergent) direct = false;
    }
#ifdef QR_WAVETIME
    direct = false;
#endif
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(scenes[0]->device));
    if (!direct)
    {
        /* no combined kernel for this set: one launch per target */
        for (int i = 0; i < n; i++)
        {
            qr_device_scene *s = scenes[i];
            const DevScene keep = s->sc;
            int rc = qr_scene_set_rows(s, row_begin[i], row_end[i], 0, 1);
            if (rc == QR_OK) { hipError_t e = launch<false>(s, frames_dev[i], nullptr, st); if (e != hipSuccess) rc = qr_fail(QR_ERR_DEVICE, hipGetErrorString(e)); }
            s->sc = keep;
            if (rc != QR_OK) return rc;
        }
        return QR_OK;
    }
    std::vector<int64_t> key;
    for (int i = 0; i < n; i++) { key.push_back((int64_t)(intptr_t)scenes[i]); key.push_back(row_begin[i]); key.push_back(row_end[i]); }
    MultiSched ms;
    {
        std::lock_guard<std::mutex> lk(g_multi_lock);
        auto it = g_multi.find(key);
        if (it == g_multi.end())
        {
            std::vector<uint32_t> ent[4];                    /* by heaviness (schedule word bits 30-31) */
            std::vector<DevScene> dsc((size_t)n);
            for (int i = 0; i < n; i++)
            {
                const qr_device_scene *s = scenes[i];
                dsc[i] = s->sc;
                dsc[i].row_begin = 0; dsc[i].row_end = s->fr.frm_h; dsc[i].index = 0; dsc[i].thnum = 1;
                dsc[i].group_first = 0; dsc[i].group_stride = 1;
                const int fh = s->fr.fsaa == 0 ? 8 : 4;
                for (size_t k = 0; k + 1 < s->h_order.size(); k += 2)
                {
                    const int y0 = (int)((s->h_order[k] >> 14) & 0x3FFFu) * fh;
                    if (y0 + fh <= row_begin[i] || y0 >= row_end[i]) continue;
                    std::vector<uint32_t> &v = ent[3 - (s->h_order[k] >> 30)];
                    v.push_back(s->h_order[k]); v.push_back(s->h_order[k + 1]); v.push_back((uint32_t)i); v.push_back(0u);
                }
            }
            std::vector<uint32_t> all;
            for (int h = 0; h < 4; h++) all.insert(all.end(), ent[h].begin(), ent[h].end());
            if (g_multi.size() >= 64)
            {
                (void)hipDeviceSynchronize();
                for (auto &kv : g_multi) { (void)hipFree(kv.second.d_order); (void)hipFree(kv.second.d_scenes); }
                g_multi.clear();
            }
            MultiSched m = { nullptr, nullptr, (int32_t)(all.size() / 4) };
            HIP_TRY(hipMalloc((void **)&m.d_scenes, dsc.size() * sizeof(DevScene)));
            HIP_TRY(hipMemcpy(m.d_scenes, dsc.data(), dsc.size() * sizeof(DevScene), hipMemcpyHostToDevice));
            if (m.n > 0)
            {
                HIP_TRY(hipMalloc((void **)&m.d_order, all.size() * 4));
                HIP_TRY(hipMemcpy(m.d_order, all.data(), all.size() * 4, hipMemcpyHostToDevice));
            }
            it = g_multi.emplace(key, m).first;
        }
        ms = it->second;
    }
    if (ms.n == 0) return QR_OK;
    DevTargets tg;
    memset(&tg, 0, sizeof(tg));
    for (int i = 0; i < n; i++)
    {
        tg.t[i].frame = (uint32_t *)frames_dev[i];
        tg.t[i].row_begin = row_begin[i]; tg.t[i].row_end = row_end[i];
        tg.t[i].scene = i;
    }
    const dim3 grid((unsigned)((ms.n + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64)), 1, 1);
    hipLaunchKernelGGL((qr_render_multi_kernel<QR_MIN_WAVES_PER_SIMD>), grid, dim3(QR_BLOCK), 0, st,
                       (const DevScene *)ms.d_scenes, tg, (const uint32_t *)ms.d_order, ms.n, scenes[0]->d_counters);
    HIP_TRY(hipGetLastError());
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
    if (s->ev0 == nullptr) { HIP_TRY(hipEventCreate(&s->ev0)); HIP_TRY(hipEventCreate(&s->ev1)); }
    for (int i = 0; i < iters; i++)
    {
#ifdef QR_WAVETIME
        HIP_TRY(hipMemsetAsync(s->d_counters + 32, 0, (size_t)s->sc.n_blocks * QR_WT_SLOTS * sizeof(unsigned long long), st));
#endif
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
        std::vector<unsigned long long> w((size_t)s->sc.n_blocks * QR_WT_SLOTS);
        HIP_TRY(hipMemcpy(w.data(), s->d_counters + 32, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        FILE *f = fopen(path, "wb");
        if (f) { fwrite(w.data(), sizeof(unsigned long long), w.size(), f); fclose(f); }
    }
#endif
    return QR_OK;
}

/* per-thread device frame + pinned staging buffer, kept between calls: the drop-in path renders a frame
 * per call, and hipMalloc/hipFree plus a pageable 8 MB copy cost more than the kernel itself */
#define QR_COPY_CHUNKS 4

/* copy into the caller's frame with non-temporal stores: the frame is written once and not read by us, so
 * read-for-ownership traffic and cache pollution of a plain memcpy are avoided (8 MB per frame) */
static void copy_streaming(void *dst, const void *src, size_t n)
{
#if defined(__SSE2__)
    uint8_t *d = (uint8_t *)dst; const uint8_t *s = (const uint8_t *)src;
    const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    if (n < 4096 || head >= n) { memcpy(dst, src, n); return; }
    memcpy(d, s, head); d += head; s += head; n -= head;
    const size_t blocks = n / 64;
    for (size_t i = 0; i < blocks; i++)
    {
        const __m128i a = _mm_loadu_si128((const __m128i *)(s + 0)), b = _mm_loadu_si128((const __m128i *)(s + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(s + 32)), e = _mm_loadu_si128((const __m128i *)(s + 48));
        _mm_stream_si128((__m128i *)(d + 0), a); _mm_stream_si128((__m128i *)(d + 16), b);
        _mm_stream_si128((__m128i *)(d + 32), c); _mm_stream_si128((__m128i *)(d + 48), e);
        s += 64; d += 64;
    }
    _mm_sfence();
    memcpy(d, s, n - blocks * 64);
#else
    memcpy(dst, src, n);
#endif
}

struct HostPathCache
{
    int device = -1;
    void *d_frame = nullptr; size_t d_cap = 0;
    uint32_t *h_frame = nullptr; size_t h_cap = 0;
    hipEvent_t ev[QR_COPY_CHUNKS] = {};
    ~HostPathCache()
    {
        /* the HIP runtime may already be shut down when thread-locals are destroyed at exit: leak */
    }
};
static thread_local HostPathCache g_hpc;

extern "C" int qr_render_host(qr_device_scene *s, uint32_t *frame_host, int row_pixels)
{
    if (s == nullptr || frame_host == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    const int w = s->fr.frm_w, h = s->fr.frm_h;
    const size_t bytes = (size_t)w * h * 4;
    HIP_TRY(hipSetDevice(s->device));
    HostPathCache &c = g_hpc;
    if (c.device != s->device || c.d_cap < bytes)
    {
        if (c.d_frame) { (void)hipSetDevice(c.device); (void)hipFree(c.d_frame); (void)hipSetDevice(s->device); }
        c.d_frame = nullptr; c.d_cap = 0; c.device = s->device;
        HIP_TRY(hipMalloc(&c.d_frame, bytes));
        c.d_cap = bytes;
    }
    if (c.h_cap < bytes)
    {
        if (c.h_frame) (void)hipHostFree(c.h_frame);
        c.h_frame = nullptr; c.h_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c.h_frame, bytes, hipHostMallocDefault));
        c.h_cap = bytes;
    }
    const bool whole = s->sc.row_begin == 0 && s->sc.row_end == h && s->sc.group_first == 0 && s->sc.group_stride == 1 && s->sc.thnum <= 1;
    hipError_t e = launch<false>(s, c.d_frame, nullptr, nullptr);
    if (e == hipSuccess && whole && row_pixels == w && h >= 64)
    {
        /* compact whole frame: the copy back runs in QR_COPY_CHUNKS row bands, the DMA of band k+1 under
         * the host memcpy of band k into the caller's frame */
        if (c.ev[0] == nullptr) for (int k = 0; k < QR_COPY_CHUNKS; k++) if (hipEventCreateWithFlags(&c.ev[k], hipEventDisableTiming) != hipSuccess) c.ev[k] = nullptr;
        bool ok = true;
        for (int k = 0; k < QR_COPY_CHUNKS; k++) ok = ok && c.ev[k] != nullptr;
        if (ok)
        {
            int y0[QR_COPY_CHUNKS + 1];
            for (int k = 0; k <= QR_COPY_CHUNKS; k++) y0[k] = (int)((long long)h * k / QR_COPY_CHUNKS);
            for (int k = 0; k < QR_COPY_CHUNKS && e == hipSuccess; k++)
            {
                const size_t off = (size_t)y0[k] * w, n = (size_t)(y0[k + 1] - y0[k]) * w * 4;
                e = hipMemcpyAsync(c.h_frame + off, (const uint32_t *)c.d_frame + off, n, hipMemcpyDeviceToHost, nullptr);
                if (e == hipSuccess) e = hipEventRecord(c.ev[k], nullptr);
            }
            for (int k = 0; k < QR_COPY_CHUNKS && e == hipSuccess; k++)
            {
                e = hipEventSynchronize(c.ev[k]);
                const size_t off = (size_t)y0[k] * w, n = (size_t)(y0[k + 1] - y0[k]) * w * 4;
                if (e == hipSuccess) copy_streaming(frame_host + off, c.h_frame + off, n);
            }
            if (e != hipSuccess) return qr_fail(QR_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
            return QR_OK;
        }
    }
    if (e == hipSuccess) e = hipMemcpy(c.h_frame, c.d_frame, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return qr_fail(QR_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
    /* copy only the rows this call owns, honouring a negative stride (bottom-up
     * frames, engine.cpp:2814-2850) */
    if (whole && row_pixels == w)
    {
        memcpy(frame_host, c.h_frame, bytes);
        return QR_OK;
    }
    for (int y = s->sc.row_begin; y < s->sc.row_end; y++)
    {
        if ((y / 8 - s->sc.group_first) % s->sc.group_stride != 0 || y / 8 < s->sc.group_first) continue;
        if (s->sc.thnum > 1 && (y % s->sc.thnum) != s->sc.index) continue;
        memcpy(frame_host + (ptrdiff_t)y * row_pixels, c.h_frame + (size_t)y * w, (size_t)w * 4);
    }
    return QR_OK;
}

/*
 * The reference entry point.  One call = flatten + upload + launch + copy back.
 * The scene is re-flattened every call because the engine rebuilds its lists
 * and animates objects every frame (engine.cpp:2976-3332); all of that is a few
 * hundred KB.
 */
extern "C" int qr_render0(const void *s_inf, const qr_abi_desc *abi)
{
    const bool verbose = getenv("QR_VERBOSE") != nullptr;
    const double t0 = now_ms();
    std::vector<uint8_t> blob;
    std::string err;
    int rc = qr_flatten_impl(s_inf, abi, blob, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    const double t1 = now_ms();

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
    const double t2 = now_ms();
    rc = qr_render_host(scn, (uint32_t *)(uintptr_t)p_frame, (int)row);
    const double t3 = now_ms();
    qr_scene_destroy(scn);
    if (verbose)
        fprintf(stderr, "qr_render0: flatten %.3f ms (%zu bytes), upload %.3f ms, render+copy %.3f ms, destroy %.3f ms\n",
                t1 - t0, blob.size(), t2 - t1, t3 - t2, now_ms() - t3);
    return rc;
}
