/*
 * qr_binning.hpp - GPU tile binning pass (QR_UPLOAD_REBIN_TILES): rebuilds the per-tile surface lists from
 * the camera list.  Host part: conservative tile rectangle per surface (double precision, worker threads); device
 * part: qr_bin_group_kernel (entries per 16x16 group of tiles) + qr_bin_kernel (one thread per tile over its group's
 * entries), each count + fill.  Included by qr_device.hip only.
 */
#ifndef QR_BINNING_HPP
#define QR_BINNING_HPP

/* ------------------------------------------------------------------------ */
/* GPU tile binning (replaces the reference's host tiling, engine.cpp:1956-2128, 3129-3253)     */
/* ------------------------------------------------------------------------ */

struct BinEntry
{
    int32_t simd;           /* surface index                                              */
    int32_t marker;         /* 1: trnode marker of an array, 0: surface                   */
    int32_t end;            /* marker: index of the last entry of its sub-list            */
    int32_t data;           /* surface: the camera-list cell's data field                 */
    int32_t x0, y0, x1, y1; /* surface: inclusive tile rectangle, x1 < x0 = off screen    */
};

#define QR_BIN_DEPTH 4      /* nesting of transformed arrays the binning kernel tracks    */
#define QR_BIN_GROUP 16     /* a group of tiles: QR_BIN_GROUP x QR_BIN_GROUP, one workgroup of the kernels below */

/*
 * Two levels, so that a tile only looks at what lies near it (one thread per tile over EVERY camera-list entry was
 * 129 600 tiles x 10 001 entries on config 5):
 *   qr_bin_group_kernel  one workgroup per group of 16x16 tiles: the entries whose rectangle meets the group (and every array
 *                        marker), in camera-list order -- 256 entries per step, kept ones compacted by ballot + prefix;
 *                        FILL = false only counts;
 *   qr_bin_kernel        one workgroup per group, one thread per tile: walks the group's entries (wave-uniform loads) and
 *                        emits the cells of the surfaces whose rectangle covers the tile; a trnode marker is emitted in front of
 *                        the first covered member of its array and its data field is patched to the last one, exactly the
 *                        structure of the engine's tile lists.  FILL = false only counts.
 */
template <bool FILL>
__global__ __launch_bounds__(256)
void qr_bin_group_kernel(const BinEntry *__restrict__ ent, int n_ent, int groups_row, int32_t *__restrict__ count,
                         const int32_t *__restrict__ offset, int32_t *__restrict__ cand)
{
    __shared__ int wave_n[4];
    __shared__ int run_base;
    const int g = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int gx0 = (g % groups_row) * QR_BIN_GROUP, gy0 = (g / groups_row) * QR_BIN_GROUP;
    const int gx1 = gx0 + QR_BIN_GROUP - 1, gy1 = gy0 + QR_BIN_GROUP - 1;
    if (tid == 0) run_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n_ent; k0 += 256)
    {
        const int k = k0 + tid;
        bool keep = false;
        if (k < n_ent)
        {
            const BinEntry e = ent[k];
            keep = e.marker != 0 || !(e.x1 < gx0 || e.x0 > gx1 || e.y1 < gy0 || e.y0 > gy1 || e.x1 < e.x0);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_n[wv] = __popcll(m);
        __syncthreads();
        int before = run_base;
        for (int w = 0; w < wv; w++) before += wave_n[w];
        if (FILL && keep) cand[offset[g] + before + __popcll(m & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        if (tid == 0) run_base += wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
        __syncthreads();
    }
    if (!FILL && tid == 0) count[g] = run_base;
}

template <bool FILL>
__global__ __launch_bounds__(256)
void qr_bin_kernel(const BinEntry *__restrict__ ent, int n_ent, int tls_row, int tls_col, int groups_row,
                   const int32_t *__restrict__ cand, const int32_t *__restrict__ cand_off,
                   int32_t *__restrict__ count, const int32_t *__restrict__ offset,
                   qr_elem *__restrict__ cells, int32_t *__restrict__ heads, int cell_base)
{
    const int g = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int tx = (g % groups_row) * QR_BIN_GROUP + (tid % QR_BIN_GROUP), ty = (g / groups_row) * QR_BIN_GROUP + (tid / QR_BIN_GROUP);
    if (tx >= tls_row || ty >= tls_col) return;
    const int t = ty * tls_row + tx;
    const int c0 = cand_off[g], c1 = cand_off[g + 1];
    const int base = FILL ? offset[t] : 0;
    int n = 0, prev = -1, sd = 0;
    int m_ent[QR_BIN_DEPTH], m_slot[QR_BIN_DEPTH], m_last[QR_BIN_DEPTH];
#pragma unroll
    for (int d = 0; d < QR_BIN_DEPTH; d++) { m_ent[d] = -1; m_slot[d] = -1; m_last[d] = -1; }

    for (int ci = c0; ci <= c1; ci++)
    {
        const int k = ci < c1 ? cand[ci] : n_ent;
        /* close the arrays that ended before entry k */
#pragma unroll
        for (int d = QR_BIN_DEPTH - 1; d >= 0; d--)
            if (d < sd && (k == n_ent || k > ent[m_ent[d]].end))
            {
                if (FILL && m_slot[d] >= 0) cells[m_slot[d]].data = cell_base + m_last[d];
                m_slot[d] = -1; sd = d;
            }
        if (k == n_ent) break;
        const BinEntry e = ent[k];
        if (e.marker)
        {
#pragma unroll
            for (int d = 0; d < QR_BIN_DEPTH; d++) if (d == sd) { m_ent[d] = k; m_slot[d] = -1; m_last[d] = -1; }
            sd++;
            continue;
        }
        if (tx < e.x0 || tx > e.x1 || ty < e.y0 || ty > e.y1) continue;
#pragma unroll
        for (int d = 0; d < QR_BIN_DEPTH; d++)
            if (d < sd && m_slot[d] < 0)
            {
                const int slot = base + n;
                if (FILL)
                {
                    qr_elem c; c.simd = ent[m_ent[d]].simd; c.data = QR_NULL; c.next = QR_NULL; c.kind = 0;
                    cells[slot] = c;
                    if (prev >= 0) cells[prev].next = cell_base + slot;
                }
                m_slot[d] = slot; prev = slot; n++;
            }
        {
            const int slot = base + n;
            if (FILL)
            {
                qr_elem c; c.simd = e.simd; c.data = e.data; c.next = QR_NULL; c.kind = 0;
                cells[slot] = c;
                if (prev >= 0) cells[prev].next = cell_base + slot;
            }
            prev = slot; n++;
#pragma unroll
            for (int d = 0; d < QR_BIN_DEPTH; d++) if (d < sd) m_last[d] = slot;
        }
    }
    if (FILL) heads[t] = n ? cell_base + base : QR_NULL;
    else count[t] = n;
}

/* conservative inclusive pixel interval [lo, hi] in which a primary ray can meet the disc that a
 * sphere projects to in the plane spanned by one image axis and the view axis; false = never */
static bool screen_interval(double a, double z, double R, double c_pix, double pov_over_step, double *lo, double *hi)
{
    const double HALF_PI = 1.5707963267948966;
    const double rho = __builtin_sqrt(a * a + z * z);
    *lo = -1e300; *hi = 1e300;
    if (!(rho > R * 1.001 + 1e-9)) return true;                 /* the eye is inside the disc */
    const double th = __builtin_atan2(a, z), al = __builtin_asin(R / rho) + 1e-6;
    const double l = th - al, h = th + al;
    if (l >= HALF_PI - 1e-6 || h <= -HALF_PI + 1e-6) return false;    /* entirely behind the image plane */
    if (l > -HALF_PI + 1e-6) *lo = c_pix + pov_over_step * __builtin_tan(l);
    if (h < HALF_PI - 1e-6) *hi = c_pix + pov_over_step * __builtin_tan(h);
    return true;
}

/*
 * Build tile lists for `frm` on the GPU from the camera list.  Appends the new cells to E and
 * fills T (frm.tls_row * frm.tls_col heads).
 */
static int rebin_tiles(const qr_scene_view &v, const std::vector<BSphere> &bsph, qr_frame &frm,
                       std::vector<qr_elem> &E, std::vector<int32_t> &T)
{
    if ((int)v.hdr->n_tiles <= 1) { frm.tile_w = 32; frm.tile_h = 8; }    /* RT_TILE_W, RT_TILE_H: engine.h:38-39 */
    if (const char *ts = getenv("QR_BIN_TILE")) { int w = 0, h = 0; if (sscanf(ts, "%dx%d", &w, &h) == 2 && w > 0 && h > 0) { frm.tile_w = w; frm.tile_h = h; } }
    frm.tls_row = (frm.frm_w + frm.tile_w - 1) / frm.tile_w;
    frm.tls_col = (frm.frm_h + frm.tile_h - 1) / frm.tile_h;
    const int n_tiles = frm.tls_row * frm.tls_col;
    const double bin_t0 = now_ms();

    /* camera model of tracer.cpp:1287-1322: ray(x, y) = dir + hor * x + ver * y from org */
    double u[3], w2[3], ww[3], hl = 0.0, vl = 0.0;
    for (int k = 0; k < 3; k++) { hl += (double)frm.hor[k] * frm.hor[k]; vl += (double)frm.ver[k] * frm.ver[k]; }
    hl = __builtin_sqrt(hl); vl = __builtin_sqrt(vl);
    bool cam_ok = hl > 0.0 && vl > 0.0;
    double hv = 0.0;
    if (cam_ok)
    {
        for (int k = 0; k < 3; k++) { u[k] = frm.hor[k] / hl; w2[k] = frm.ver[k] / vl; hv += u[k] * w2[k]; }
        ww[0] = u[1] * w2[2] - u[2] * w2[1]; ww[1] = u[2] * w2[0] - u[0] * w2[2]; ww[2] = u[0] * w2[1] - u[1] * w2[0];
        if (hv > 1e-6 || hv < -1e-6) cam_ok = false;           /* skewed image axes: fall back to full-screen rectangles */
    }
    double pov = 0.0, cx = 0.0, cy = 0.0;
    if (cam_ok)
    {
        double du = 0.0, dv = 0.0;
        for (int k = 0; k < 3; k++) { pov += frm.dir[k] * ww[k]; du += frm.dir[k] * u[k]; dv += frm.dir[k] * w2[k]; }
        if (pov < 0.0) { pov = -pov; for (int k = 0; k < 3; k++) ww[k] = -ww[k]; }
        if (!(pov > 1e-9)) cam_ok = false;
        cx = -du / hl; cy = -dv / vl;
    }

    /* the camera list's cells, then the rectangle of every surface among them (worker threads: the eight corners of a box
     * clipped against the view pyramid in double precision are 0.7 us a surface), then the entries in camera-list order */
    std::vector<int> cl;
    for (int c = frm.clist; c != QR_NULL; c = E[c].next)
    {
        if (cl.size() > E.size()) return qr_fail(QR_ERR_ARG, "cyclic camera list");
        cl.push_back(c);
    }
    std::vector<BinEntry> rects(cl.size());
    auto rect_range = [&](size_t i0, size_t i1) {
        for (size_t ri = i0; ri < i1; ri++)
        {
            const qr_elem &el = E[cl[ri]];
            const qr_surface &q = v.srf[el.simd];
            BinEntry b; memset(&b, 0, sizeof(b));
            b.simd = el.simd; b.data = el.data;
            if ((el.kind & 3) != 1 && q.srf_t[3] >= 0)
            {
            const BSphere &bs = bsph[el.simd];
                b.x0 = 0; b.y0 = 0; b.x1 = frm.tls_row - 1; b.y1 = frm.tls_col - 1;
                static const int diag = []() { const char *e = getenv("QR_BIN_DIAG"); return e ? atoi(e) : 0; }();   /* diagnosis: 1 transformed surfaces cover the screen, 2 all do */
                const bool full = diag == 2 || (diag == 1 && (q.has_trm != 0 || q.shift != 0));
                if (cam_ok && bs.r < 1e30f && !full)
                {
                    const double R = (double)bs.r * 1.001 + 1e-6;
                    double a = 0.0, bb = 0.0, z = 0.0;
                    for (int k = 0; k < 3; k++) { const double qk = (double)bs.c[k] - frm.org[k]; a += qk * u[k]; bb += qk * w2[k]; z += qk * ww[k]; }
                    double xl, xh, yl, yh;
                    const bool vx = screen_interval(a, z, R, cx, pov / hl, &xl, &xh);
                    const bool vy = screen_interval(bb, z, R, cy, pov / vl, &yl, &yh);
                    /* tighter: the projected corners of the surface's bounding box, when all of them lie in
                     * front of the image plane (planes and clipped shapes fill their box far better than
                     * their sphere) */
                    BBox bx;
                    (void)bound_sphere(v, el.simd, &bx);
                    static const int box_mode = []() { const char *e = getenv("QR_BIN_BOX"); return e ? atoi(e) : 2; }();
                    /* box_mode (QR_BIN_BOX, diagnosis): 0 spheres only; 1 boxes of surfaces without scaling or transform node; 2 all (default) */
                    const bool box_ok = box_mode == 2 || (box_mode == 1 && q.has_trm == 0 && q.shift == 0);
                    if (bx.valid && vx && vy && box_ok)
                    {
                        /* camera-space corners; the part of the box in front of the plane z = zn is the convex hull
                         * of the corners in front and of the points where box edges cross that plane, so the
                         * rectangle of their projections bounds everything a primary ray can meet */
                        double cam[8][3];
                        for (int c = 0; c < 8; c++)
                        {
                            double pa = 0.0, pb = 0.0, pz = 0.0;
                            for (int k = 0; k < 3; k++) { const double qk = bx.p[c][k] - frm.org[k]; pa += qk * u[k]; pb += qk * w2[k]; pz += qk * ww[k]; }
                            cam[c][0] = pa; cam[c][1] = pb; cam[c][2] = pz;
                        }
                        const double zn = 1e-3 * pov;
                        double bxl = 1e300, bxh = -1e300, byl = 1e300, byh = -1e300; int npts = 0;
                        /* Each face of the box is clipped (Sutherland-Hodgman) against the eye plane and the four
                         * sides of the view pyramid widened by 4 pixels, and what is left is projected: a wall that
                         * runs past the eye then covers the part of the screen it fills, not all of it. */
                        const double kx0 = (-4.0 - cx) * hl / pov, kx1 = (frm.frm_w + 4.0 - cx) * hl / pov;
                        const double ky0 = (-4.0 - cy) * vl / pov, ky1 = (frm.frm_h + 4.0 - cy) * vl / pov;
                        static const int face[6][4] = { {0, 1, 3, 2}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 3, 7, 5} };
                        for (int f = 0; f < 6; f++)
                        {
                            double poly[16][3], tmp[16][3]; int np = 4;
                            for (int i = 0; i < 4; i++) for (int k = 0; k < 3; k++) poly[i][k] = cam[face[f][i]][k];
                            for (int pl = 0; pl < 5 && np > 0; pl++)
                            {
                                auto dist = [&](const double *q) {
                                    switch (pl) {
                                    case 0: return q[2] - zn;
                                    case 1: return q[0] - kx0 * q[2];
                                    case 2: return kx1 * q[2] - q[0];
                                    case 3: return q[1] - ky0 * q[2];
                                    default: return ky1 * q[2] - q[1];
                                    }
                                };
                                int nt = 0;
                                for (int i = 0; i < np; i++)
                                {
                                    const double *p0 = poly[i], *p1 = poly[(i + 1) % np];
                                    const double d0 = dist(p0), d1 = dist(p1);
                                    if (d0 >= 0.0) { for (int k = 0; k < 3; k++) tmp[nt][k] = p0[k]; nt++; }
                                    if ((d0 >= 0.0) != (d1 >= 0.0))
                                    {
                                        const double t = d0 / (d0 - d1);
                                        for (int k = 0; k < 3; k++) tmp[nt][k] = p0[k] + t * (p1[k] - p0[k]);
                                        nt++;
                                    }
                                }
                                np = nt < 16 ? nt : 16;
                                for (int i = 0; i < np; i++) for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
                            }
                            for (int i = 0; i < np; i++)
                            {
                                const double pz = poly[i][2] > zn ? poly[i][2] : zn;
                                const double sx = cx + (poly[i][0] / pz) * (pov / hl), sy = cy + (poly[i][1] / pz) * (pov / vl);
                                if (sx < bxl) bxl = sx; if (sx > bxh) bxh = sx; if (sy < byl) byl = sy; if (sy > byh) byh = sy;
                                npts++;
                            }
                        }
                        if (npts == 0) { xl = 1e300; xh = -1e300; }           /* the whole box is behind the eye */
                        else
                        {
                            /* a margin relative to the box size covers the 1e-3 inflation of the bounds */
                            const double ex = 2e-3 * (bxh - bxl) + 1e-3, ey = 2e-3 * (byh - byl) + 1e-3;
                            if (bxl - ex > xl) xl = bxl - ex; if (bxh + ex < xh) xh = bxh + ex;
                            if (byl - ey > yl) yl = byl - ey; if (byh + ey < yh) yh = byh + ey;
                        }
                    }
                    const double mg = 2.0;                              /* FSAA sample offsets (< 0.5 px) + fp32 ray rounding */
                    if (!vx || !vy || xl > xh || yl > yh || xh + mg < 0.0 || yh + mg < 0.0 || xl - mg > frm.frm_w || yl - mg > frm.frm_h) { b.x0 = 1; b.x1 = 0; }
                    else
                    {
                        auto tl = [](double p, int ts, int nt, bool up) { double t = __builtin_floor(p / ts); if (t < 0) t = 0; if (t > nt - 1) t = nt - 1; (void)up; return (int32_t)t; };
                        b.x0 = tl(xl - mg, frm.tile_w, frm.tls_row, false); b.x1 = tl(xh + mg, frm.tile_w, frm.tls_row, true);
                        b.y0 = tl(yl - mg, frm.tile_h, frm.tls_col, false); b.y1 = tl(yh + mg, frm.tile_h, frm.tls_col, true);
                    }
                }
            }
            rects[ri] = b;
        }
    };
    {
        const char *te = getenv("QR_HOST_THREADS");
        int n_thr = te ? atoi(te) : (int)std::thread::hardware_concurrency();
        if (n_thr > 16) n_thr = 16;
        if ((size_t)n_thr > cl.size() / 256) n_thr = (int)(cl.size() / 256);
        if (n_thr <= 1) rect_range(0, cl.size());
        else
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < n_thr; t++) pool.emplace_back(rect_range, cl.size() * (size_t)t / (size_t)n_thr, cl.size() * (size_t)(t + 1) / (size_t)n_thr);
            for (std::thread &t : pool) t.join();
        }
    }
    std::vector<BinEntry> ent;
    std::vector<int> open_end_cell;                             /* stack of the cells that end the open arrays */
    std::vector<int> open_entry;
    for (size_t ri = 0; ri < cl.size(); ri++)
    {
        const int c = cl[ri];
        const qr_elem &el = E[c];
        const qr_surface &q = v.srf[el.simd];
        BinEntry b = rects[ri];
        bool emit = true;
        if ((el.kind & 3) == 1) emit = false;                   /* bounding-volume cell: tile lists carry none (engine.cpp:1711-1725) */
        else if (q.srf_t[3] < 0)
        {
            b.marker = 1; b.end = -1;
            if ((int)open_entry.size() >= QR_BIN_DEPTH) return qr_fail(QR_ERR_UNSUP, "transformed arrays nested deeper than the tile binning supports");
        }
        if (emit)
        {
            if (b.marker) { open_entry.push_back((int)ent.size()); open_end_cell.push_back(el.data); }
            ent.push_back(b);
        }
        /* close the arrays whose last cell this is (also when that cell itself was dropped) */
        while (!open_entry.empty() && open_end_cell.back() == c)
        {
            ent[open_entry.back()].end = (int32_t)ent.size() - 1;
            open_entry.pop_back(); open_end_cell.pop_back();
        }
    }
    if (!open_entry.empty()) return qr_fail(QR_ERR_ARG, "array in the camera list does not end inside the list");

    T.assign((size_t)n_tiles, QR_NULL);
    if (ent.empty()) return QR_OK;
    const bool bin_verbose = getenv("QR_VERBOSE") && atoi(getenv("QR_VERBOSE")) >= 2;
    double bin_t = bin_t0;
    auto bin_phase = [&](const char *name) { if (bin_verbose) { const double t = now_ms(); fprintf(stderr, "  binning %-10s %.3f ms\n", name, t - bin_t); bin_t = t; } };
    bin_phase("rectangles");
    if (const char *vb = getenv("QR_VERBOSE"))
        if (atoi(vb) >= 3)
            for (const BinEntry &b : ent)
                if (!b.marker)
                    fprintf(stderr, "bin entry: surface %d tiles %d (x %d..%d, y %d..%d)\n", b.simd,
                            b.x1 < b.x0 ? 0 : (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1), b.x0, b.x1, b.y0, b.y1);

    BinEntry *d_ent = nullptr; int32_t *d_cnt = nullptr, *d_off = nullptr, *d_heads = nullptr, *d_gcnt = nullptr, *d_goff = nullptr, *d_cand = nullptr;
    qr_elem *d_cells = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_ent); (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_heads); (void)hipFree(d_cells);
                           (void)hipFree(d_gcnt); (void)hipFree(d_goff); (void)hipFree(d_cand); };
#define BIN_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return qr_fail(QR_ERR_DEVICE, std::string(#x ": ") + hipGetErrorString(e_)); } } while (0)
    const int groups_row = (frm.tls_row + QR_BIN_GROUP - 1) / QR_BIN_GROUP, groups_col = (frm.tls_col + QR_BIN_GROUP - 1) / QR_BIN_GROUP;
    const int n_groups = groups_row * groups_col;
    BIN_TRY(hipMalloc((void **)&d_ent, ent.size() * sizeof(BinEntry)));
    BIN_TRY(hipMalloc((void **)&d_cnt, (size_t)n_tiles * 4));
    BIN_TRY(hipMalloc((void **)&d_off, (size_t)n_tiles * 4));
    BIN_TRY(hipMalloc((void **)&d_heads, (size_t)n_tiles * 4));
    BIN_TRY(hipMalloc((void **)&d_gcnt, (size_t)n_groups * 4));
    BIN_TRY(hipMalloc((void **)&d_goff, ((size_t)n_groups + 1) * 4));
    BIN_TRY(hipMemcpy(d_ent, ent.data(), ent.size() * sizeof(BinEntry), hipMemcpyHostToDevice));
    bin_phase("alloc+h2d");
    const dim3 blk(256), grd((unsigned)n_groups);
    /* level 1: the entries of every group of tiles */
    hipLaunchKernelGGL((qr_bin_group_kernel<false>), grd, blk, 0, 0, d_ent, (int)ent.size(), groups_row, d_gcnt, (const int32_t *)nullptr, (int32_t *)nullptr);
    BIN_TRY(hipGetLastError());
    std::vector<int32_t> gcnt((size_t)n_groups), goff((size_t)n_groups + 1);
    BIN_TRY(hipMemcpy(gcnt.data(), d_gcnt, (size_t)n_groups * 4, hipMemcpyDeviceToHost));
    uint64_t n_cand = 0;
    for (int i = 0; i < n_groups; i++) { goff[i] = (int32_t)n_cand; n_cand += (uint32_t)gcnt[i]; }
    goff[n_groups] = (int32_t)n_cand;
    if (n_cand > 0x7FFFFFF0ull) { cleanup(); return qr_fail(QR_ERR_NOMEM, "tile binning: too many (group, entry) pairs"); }
    BIN_TRY(hipMalloc((void **)&d_cand, (size_t)(n_cand ? n_cand : 1) * 4));
    BIN_TRY(hipMemcpy(d_goff, goff.data(), ((size_t)n_groups + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((qr_bin_group_kernel<true>), grd, blk, 0, 0, d_ent, (int)ent.size(), groups_row, d_gcnt, (const int32_t *)d_goff, d_cand);
    BIN_TRY(hipGetLastError());
    bin_phase("groups");
    /* level 2: the cells of every tile */
    hipLaunchKernelGGL((qr_bin_kernel<false>), grd, blk, 0, 0, d_ent, (int)ent.size(), frm.tls_row, frm.tls_col, groups_row, (const int32_t *)d_cand, (const int32_t *)d_goff,
                       d_cnt, (const int32_t *)nullptr, (qr_elem *)nullptr, (int32_t *)nullptr, 0);
    BIN_TRY(hipGetLastError());
    std::vector<int32_t> cnt((size_t)n_tiles), off((size_t)n_tiles);
    BIN_TRY(hipMemcpy(cnt.data(), d_cnt, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    bin_phase("count");
    uint64_t total = 0;
    for (int i = 0; i < n_tiles; i++) { off[i] = (int32_t)total; total += (uint32_t)cnt[i]; }
    if (total + E.size() > 0x7FFFFFF0ull) { cleanup(); return qr_fail(QR_ERR_NOMEM, "tile lists exceed the 31-bit cell index space"); }
    const int cell_base = (int)E.size();
    if (total > 0)
    {
        BIN_TRY(hipMalloc((void **)&d_cells, (size_t)total * sizeof(qr_elem)));
        BIN_TRY(hipMemcpy(d_off, off.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((qr_bin_kernel<true>), grd, blk, 0, 0, d_ent, (int)ent.size(), frm.tls_row, frm.tls_col, groups_row, (const int32_t *)d_cand, (const int32_t *)d_goff,
                           d_cnt, (const int32_t *)d_off, d_cells, d_heads, cell_base);
        BIN_TRY(hipGetLastError());
        /* (callers reserve room behind their cells: growing a 32 MB element array by its 5.5 MB of tile cells would copy it --
         * 5 of the pass's 8.5 ms on config 5) */
        E.resize((size_t)cell_base + total);
        BIN_TRY(hipMemcpy(E.data() + cell_base, d_cells, (size_t)total * sizeof(qr_elem), hipMemcpyDeviceToHost));
        BIN_TRY(hipMemcpy(T.data(), d_heads, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    }
    bin_phase("fill+d2h");
    cleanup();
    bin_phase("free");
#undef BIN_TRY
    if (getenv("QR_VERBOSE"))
        fprintf(stderr, "tile binning: %zu camera-list entries x %d tiles (%dx%d px) -> %llu cells\n", ent.size(), n_tiles, frm.tile_w, frm.tile_h, (unsigned long long)total);
    return QR_OK;
}

#endif /* QR_BINNING_HPP */
