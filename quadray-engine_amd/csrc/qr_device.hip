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
#include <thread>
#include <mutex>
#include <map>
#include <atomic>
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
    uint64_t serial = 0;        /* unique per upload for the life of the process: caches keyed on a scene (the QR_DEVICES replicas of
                                 * qr_render_host) use it -- a destroyed scene's address and its image's device address are reused */
    int device = 0;
    void *d_blob = nullptr;     /* the compiled scene image (qr_program.h), one allocation */
    uint64_t blob_bytes = 0;
    LaunchP lp = {};            /* blob pointer + launch parameters          */
    qr_frame fr = {};           /* host copy of the frame parameters        */
    qr_header hdr = {};
    unsigned long long *d_counters = nullptr;
    size_t n_cells = 0;
    int32_t n_groups = 0;
    bool divk = false;          /* some list is a long hierarchy: launch the kernel instance with the per-lane walk */
    /* path-tracer mode (qr_scene_set_pt): what the engine keeps per frame buffer, engine.cpp:2875-2893 */
    bool pt_on = false;
    uint32_t *d_seeds = nullptr; float *d_acc = nullptr;     /* frm_row * frm_h * samples each; d_acc holds r, g, b planes */
    uint64_t pt_frames = 0;
    int pt_eager = 0;           /* 1: the reference's shading order (qr_pt_eager.hpp): its random streams, slow; 3: self-test */
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    /* the whole-frame wave schedule (host copy) and the schedules of the row selections rendered so far:
     * a launch restricted by qr_scene_set_rows / _set_tile_rows only starts the waves that own pixels */
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

/* Tried and dropped for the drop-in path: page-locking the engine's frame in place for a direct DMA
 * copy-back saves 0.35 ms per 1080p frame on demo1 but makes demo2 ten times slower (host accesses to the
 * engine's heap next to the frame and later copies slow down once the range is registered). */

#include "qr_bounds.hpp"
#include "qr_binning.hpp"

extern "C" int qr_scene_upload(const void *blob, uint64_t size, int device, qr_device_scene **out)
{
    return qr_scene_upload_ex(blob, size, device, 0u, out);
}

static int pick_device(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return qr_fail(QR_ERR_DEVICE, "no HIP device available (the gfx950 backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return qr_fail(QR_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return QR_OK;
}

/* host half of an upload: validate, bound, (re-bin), compile -> QrProgram */
static int build_program(const void *blob, uint64_t size, int device, uint32_t flags, QrProgram &prog, qr_header &hdr, size_t &n_cells)
{
    const bool ph_verbose = getenv("QR_VERBOSE") && atoi(getenv("QR_VERBOSE")) >= 2;
    double ph_t = now_ms();
    auto phase = [&](const char *name) { if (ph_verbose) { const double t = now_ms(); fprintf(stderr, "upload phase %-12s %.3f ms\n", name, t - ph_t); ph_t = t; } };
    if (getenv("QR_REBIN") && atoi(getenv("QR_REBIN")) != 0) flags |= QR_UPLOAD_REBIN_TILES;
    qr_scene_view v;
    int rc = qr_scene_view_init(&v, blob, size);
    if (rc != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot (qr_scene_view_init " + std::to_string(rc) + ")");
    std::string err;
    rc = qr_snapshot_validate(v, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    phase("validate");
    std::vector<BSphere> bsph;
    qr_bound_spheres(v, bsph);
    if (getenv("QR_VERBOSE"))
    {
        int nreal = 0, nfin = 0;
        for (uint32_t i = 0; i < v.hdr->n_srf; i++)
        {
            const qr_surface &q = v.srf[i];
            if (q.srf_t[3] < 0 || q.srf_t[3] >= QR_TAG_SURFACE_MAX) continue;
            nreal++; if (bsph[i].r < 1e30f) nfin++;
        }
        fprintf(stderr, "bounding spheres: %d of %d real surfaces bounded\n", nfin, nreal);
    }
    phase("bounds");
    /* working copies: the tile lists (and with them the cell array and the tile geometry of the
     * frame record) are replaced when the binning pass runs */
    std::vector<qr_elem> E;
    if (flags & QR_UPLOAD_REBIN_TILES) E.reserve((size_t)v.hdr->n_elm + (size_t)v.hdr->n_elm / 2 + 65536);     /* room for the tile cells */
    E.assign(v.elm, v.elm + v.hdr->n_elm);
    std::vector<int32_t> T(v.tiles, v.tiles + v.hdr->n_tiles);
    qr_frame frm = *v.frame;
    if (flags & QR_UPLOAD_REBIN_TILES)
    {
        rc = pick_device(device);
        if (rc != QR_OK) return rc;
        rc = rebin_tiles(v, bsph, frm, E, T);
        if (rc != QR_OK) return rc;
        phase("rebin");
    }
    rc = qr_program_build(v, E, T, frm, bsph, prog, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    phase("compile");
    hdr = *v.hdr;
    n_cells = E.size();
    return QR_OK;
}

extern "C" int qr_scene_upload_ex(const void *blob, uint64_t size, int device, uint32_t flags, qr_device_scene **out)
{
    if (blob == nullptr || out == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    QrProgram prog;
    qr_header hdr; size_t n_cells = 0;
    int rc = build_program(blob, size, device, flags, prog, hdr, n_cells);
    if (rc != QR_OK) return rc;
    rc = pick_device(device);
    if (rc != QR_OK) return rc;
    const size_t total = prog.blob.size();

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
    memcpy(stage.p, prog.blob.data(), total);

    qr_device_scene *s = new qr_device_scene();     /* value-initialised: plain members are zero */
    static std::atomic<uint64_t> next_serial{1};
    s->serial = next_serial.fetch_add(1, std::memory_order_relaxed);
    s->device = device;
    s->hdr = hdr;
    const double tm0 = now_ms();
    hipError_t e = hipMalloc(&s->d_blob, total);
    if (e != hipSuccess) { delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    const double tm1 = now_ms();
    e = hipMemcpy(s->d_blob, stage.p, total, hipMemcpyHostToDevice);
    if (getenv("QR_VERBOSE")) fprintf(stderr, "upload: device bytes %zu, hipMalloc %.3f ms, hipMemcpy %.3f ms\n", total, tm1 - tm0, now_ms() - tm1);
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    const size_t n_sched = prog.n_sched;
#ifdef QR_WAVETIME
    e = hipMalloc((void **)&s->d_counters, (64 + QR_WT_SLOTS * n_sched) * sizeof(unsigned long long));
#else
    e = hipMalloc((void **)&s->d_counters, 64 * sizeof(unsigned long long));
#endif
    if (e != hipSuccess) { (void)hipFree(s->d_blob); delete s; return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    s->blob_bytes = total;

    const qr_frame &frm = prog.frm;
    s->lp.B = (const char *)s->d_blob;
    s->lp.order = (const uint32_t *)((const char *)s->d_blob + prog.off_order);
    s->lp.n_blocks = (int32_t)n_sched;
    s->h_order.swap(prog.order);
    s->lp.stats = s->d_counters + 4;
    s->fr = frm;
    s->n_cells = n_cells;
    {
        const char *dv = getenv("QR_DIV");          /* QR_DIV=0 / 1 forces the kernel instance (experiments, tests) */
        s->divk = (dv ? atoi(dv) != 0 : prog.has_long_lists) || prog.has_grids;      /* only that instance knows shadow grids */
    }
    s->lp.depth = frm.depth > QR_MAX_DEPTH ? QR_MAX_DEPTH : frm.depth;
    s->lp.row_begin = 0; s->lp.row_end = frm.frm_h;
    s->lp.index = frm.index; s->lp.thnum = frm.thnum > 0 ? frm.thnum : 1;
    s->lp.group_first = 0; s->lp.group_stride = 1;
    s->n_groups = (frm.frm_h + 7) / 8;
    s->lp.dbg = getenv("QR_DBG") ? atoi(getenv("QR_DBG")) : 0;
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
    (void)hipFree(s->d_seeds);
    (void)hipFree(s->d_acc);
    delete s;
    return QR_OK;
}

extern "C" int qr_scene_get_info(const qr_device_scene *s, qr_scene_info *info)
{
    if (s == nullptr || info == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    memset(info, 0, sizeof(*info));
    info->frm_w = s->fr.frm_w; info->frm_h = s->fr.frm_h;
    info->fsaa = s->fr.fsaa; info->depth = s->lp.depth;
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
    s->lp.depth = depth;
    return QR_OK;
}

/*
 * Path-tracer mode on / off.  Turning it on (re)starts the accumulation: the seed plane as rt_Scene::reset_pseed
 * fills it (engine.cpp:3651-3685: a 48-bit LCG walks over the slots, each slot keeps the low 32 bits), colour planes 0.
 */
extern "C" int qr_scene_set_pt(qr_device_scene *s, int on)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    if (!on) { s->pt_on = false; return QR_OK; }
    const size_t n = (size_t)s->fr.frm_row * s->fr.frm_h * ((size_t)1 << s->fr.fsaa);
    if (s->fr.frm_row < s->fr.frm_w || n == 0 || n > ((size_t)1 << 30)) return qr_fail(QR_ERR_ARG, "bad frame stride for the sample planes");
    if (s->d_seeds == nullptr || s->d_acc == nullptr)
    {
        /* both planes or neither: a half-built pair would be taken for a finished one by the next call */
        uint32_t *seeds_dev = nullptr; float *acc_dev = nullptr;
        HIP_TRY(hipMalloc((void **)&seeds_dev, n * sizeof(uint32_t)));
        const hipError_t ea = hipMalloc((void **)&acc_dev, 3 * n * sizeof(float));
        if (ea != hipSuccess) { (void)hipFree(seeds_dev); return qr_fail(QR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(ea)); }
        (void)hipFree(s->d_seeds); (void)hipFree(s->d_acc);
        s->d_seeds = seeds_dev; s->d_acc = acc_dev;
    }
    std::vector<uint32_t> seeds(n);
    unsigned long long seed = 1;
    for (size_t k = 0; k < n; k++)
    {
        seed = (seed * 25214903917ull + 11ull) & 0x0000FFFFFFFFFFFFull;
        seeds[k] = (uint32_t)seed;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s->d_seeds, seeds.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->d_acc, 0, 3 * n * sizeof(float)));
    s->pt_frames = 0;
    s->pt_on = true;
    s->pt_eager = on == 2 ? 1 : (on == 3 ? 3 : 0);
    return QR_OK;
}

extern "C" int qr_scene_set_rows(qr_device_scene *s, int row_begin, int row_end, int index, int thnum)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    const int h = s->fr.frm_h;
    if (row_begin < 0 || row_end > h || row_begin > row_end) return qr_fail(QR_ERR_ARG, "bad row range");
    if (thnum <= 0 || index < 0 || index >= thnum) return qr_fail(QR_ERR_ARG, "bad index/thnum");
    s->lp.row_begin = row_begin; s->lp.row_end = row_end;
    s->lp.index = index; s->lp.thnum = thnum;
    s->lp.group_first = row_begin / 8;
    s->lp.group_stride = 1;
    s->n_groups = row_end > row_begin ? (row_end - 1) / 8 - row_begin / 8 + 1 : 0;
    return QR_OK;
}

extern "C" int qr_scene_set_tile_rows(qr_device_scene *s, int first, int stride)
{
    if (s == nullptr) return qr_fail(QR_ERR_ARG, "null scene");
    const int total = (s->fr.frm_h + 7) / 8;
    if (stride <= 0 || first < 0) return qr_fail(QR_ERR_ARG, "bad tile-row selection");
    s->lp.row_begin = 0; s->lp.row_end = s->fr.frm_h;
    s->lp.index = 0; s->lp.thnum = 1;
    s->lp.group_first = first; s->lp.group_stride = stride;
    s->n_groups = first < total ? (total - first + stride - 1) / stride : 0;
    return QR_OK;
}

template <bool COUNT>
static hipError_t launch(qr_device_scene *s, void *frame_dev, int32_t *ids_dev, hipStream_t st)
{
    const int fsaa = s->fr.fsaa;
    if (s->n_groups == 0) return hipSuccess;
    LaunchP lp = s->lp;
    const int H = s->fr.frm_h;
    const bool whole = lp.row_begin == 0 && lp.row_end == H && lp.group_first == 0 && lp.group_stride == 1 && lp.thnum <= 1;
    if (!whole)
    {
        const std::array<int32_t, 6> key = { lp.row_begin, lp.row_end, lp.group_first, lp.group_stride, lp.index, lp.thnum };
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
                    any = y >= lp.row_begin && y < lp.row_end && g >= lp.group_first && (g - lp.group_first) % lp.group_stride == 0
                       && (lp.thnum <= 1 || (y % lp.thnum) == lp.index);
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
        lp.order = it->second.d_order;
        lp.n_blocks = it->second.n;
    }
    dim3 grid((lp.n_blocks + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64), 1, 1);
    if (grid.x == 0) return hipSuccess;
    /* register budget variant (waves per SIMD); QR_WAVES is a tuning knob for experiments */
    static const int waves = []() { const char *e = getenv("QR_WAVES"); int w = e ? atoi(e) : QR_MIN_WAVES_PER_SIMD;
                                    return (w == 3 || w == 4 || w == 5) ? w : QR_MIN_WAVES_PER_SIMD; }();
    uint32_t *f = (uint32_t *)frame_dev;
    (void)waves;
    if (s->pt_on)
    {
        /* one more sample of every pixel sample: weights of the running mean, tracer.cpp:1112-1136 */
        if (COUNT || ids_dev != nullptr) return hipErrorInvalidValue;
        s->pt_frames++;
        PtParams pt;
        const size_t n = (size_t)s->fr.frm_row * s->fr.frm_h * ((size_t)1 << s->fr.fsaa);
        pt.seeds = s->d_seeds; pt.acc_r = s->d_acc; pt.acc_g = s->d_acc + n; pt.acc_b = s->d_acc + 2 * n;
        pt.pts_o = 1.0f / (float)s->pt_frames; pt.pts_u = 1.0f - pt.pts_o;
        pt.eager = s->pt_eager; pt.pad = 0;
        hipLaunchKernelGGL(qr_render_pt_kernel, grid, dim3(QR_BLOCK), 0, st, lp, pt, f, s->d_counters);
        return hipGetLastError();
    }
    if (s->divk) hipLaunchKernelGGL((qr_render_kernel<COUNT, QR_DIVK_WAVES, true>), grid, dim3(QR_BLOCK), 0, st, lp, f, ids_dev, s->d_counters);
#ifdef QR_WAVE_VARIANTS
    else if (!COUNT && waves == 3) hipLaunchKernelGGL((qr_render_kernel<false, 3, false>), grid, dim3(QR_BLOCK), 0, st, lp, f, ids_dev, s->d_counters);
    else if (!COUNT && waves == 5) hipLaunchKernelGGL((qr_render_kernel<false, 5, false>), grid, dim3(QR_BLOCK), 0, st, lp, f, ids_dev, s->d_counters);
#endif
    else hipLaunchKernelGGL((qr_render_kernel<COUNT, 4, false>), grid, dim3(QR_BLOCK), 0, st, lp, f, ids_dev, s->d_counters);
    return hipGetLastError();
}

/* A path-traced frame is ONE more sample of every pixel sample: the sample count (and with it the weights of the running
 * mean, tracer.cpp:1112-1136) advances once per launch, so a frame cut into several row-range launches would weigh its
 * later ranges wrongly.  Such launches are refused; the drop-in entry point, where the engine itself cuts a frame into
 * index / thnum slices, takes the count from the caller's s_inf instead (dropin_pt_begin). */
static int pt_rows_ok(const qr_device_scene *s)
{
    if (!s->pt_on) return QR_OK;
    const bool whole = s->lp.row_begin == 0 && s->lp.row_end == s->fr.frm_h && s->lp.group_first == 0 && s->lp.group_stride == 1 && s->lp.thnum <= 1;
    if (!whole) return qr_fail(QR_ERR_UNSUP, "path-tracer mode renders whole frames only: reset qr_scene_set_rows / qr_scene_set_tile_rows to the full frame");
    return QR_OK;
}

extern "C" int qr_render_async(qr_device_scene *s, void *frame_dev, void *stream)
{
    if (s == nullptr || frame_dev == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    { const int rc = pt_rows_ok(s); if (rc != QR_OK) return rc; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch<false>(s, frame_dev, nullptr, (hipStream_t)stream));
    return QR_OK;
}

/* combined schedules of multi-target launches, keyed by (scene, row range) per target; only the schedule is
 * cached -- recursion depth and frame pointers travel in the kernel arguments of every launch */
struct MultiSched { uint32_t *d_order; int32_t n; };
static std::map<std::vector<int64_t>, MultiSched> g_multi;
static std::mutex g_multi_lock;

static void multi_forget(const qr_device_scene *s)
{
    std::lock_guard<std::mutex> lk(g_multi_lock);
    for (auto it = g_multi.begin(); it != g_multi.end(); )
    {
        bool uses = false;
        for (size_t i = 0; i < it->first.size(); i += 3) if (it->first[i] == (int64_t)(intptr_t)s) uses = true;
        if (uses) { (void)hipFree(it->second.d_order); it = g_multi.erase(it); }
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
        if (scenes[i]->pt_on) return qr_fail(QR_ERR_UNSUP, "a scene in path-tracer mode cannot be part of a multi-target launch");
        if (scenes[i]->device != scenes[0]->device) return qr_fail(QR_ERR_ARG, "scenes live on different devices");
        if (row_begin[i] < 0 || row_end[i] > scenes[i]->fr.frm_h || row_begin[i] > row_end[i]) return qr_fail(QR_ERR_ARG, "bad row range");
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
            const LaunchP keep = s->lp; const int32_t keep_groups = s->n_groups;
            int rc = qr_scene_set_rows(s, row_begin[i], row_end[i], 0, 1);
            if (rc == QR_OK) { hipError_t e = launch<false>(s, frames_dev[i], nullptr, st); if (e != hipSuccess) rc = qr_fail(QR_ERR_DEVICE, hipGetErrorString(e)); }
            s->lp = keep; s->n_groups = keep_groups;
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
            for (int i = 0; i < n; i++)
            {
                const qr_device_scene *s = scenes[i];
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
                for (auto &kv : g_multi) (void)hipFree(kv.second.d_order);
                g_multi.clear();
            }
            MultiSched m = { nullptr, (int32_t)(all.size() / 4) };
            if (m.n > 0)
            {
                HIP_TRY(hipMalloc((void **)&m.d_order, all.size() * 4));
                const hipError_t e = hipMemcpy(m.d_order, all.data(), all.size() * 4, hipMemcpyHostToDevice);
                if (e != hipSuccess) { (void)hipFree(m.d_order); return qr_fail(QR_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
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
        tg.t[i].B = scenes[i]->lp.B;
        tg.t[i].row_begin = row_begin[i]; tg.t[i].row_end = row_end[i];
        tg.t[i].depth = scenes[i]->lp.depth;
    }
    const dim3 grid((unsigned)((ms.n + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64)), 1, 1);
    bool divk = false;
    for (int i = 0; i < n; i++) divk = divk || scenes[i]->divk;
    if (divk) hipLaunchKernelGGL((qr_render_multi_kernel<QR_DIVK_WAVES, true>), grid, dim3(QR_BLOCK), 0, st,
                                 tg, (const uint32_t *)ms.d_order, ms.n, scenes[0]->d_counters);
    else      hipLaunchKernelGGL((qr_render_multi_kernel<QR_MIN_WAVES_PER_SIMD, false>), grid, dim3(QR_BLOCK), 0, st,
                                 tg, (const uint32_t *)ms.d_order, ms.n, scenes[0]->d_counters);
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
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, 64 * sizeof(unsigned long long), st));
    HIP_TRY(launch<true>(s, frame_dev, nullptr, st));
    unsigned long long h[4];
    HIP_TRY(hipMemcpyAsync(h, s->d_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    counts->primary = h[0]; counts->shadow = h[1]; counts->reflect = h[2]; counts->refract = h[3];
#ifdef QR_PROF
    {
        unsigned long long pf[64];
        HIP_TRY(hipMemcpyFromSymbol(pf, HIP_SYMBOL(qr_prof), sizeof(pf)));
        static const char *nm[48] = { "candidates through clip()", "clipper programs run", "clipper cells", "  fast plane cells", "  trnode / trsame cells",
            "  generic plane tests", "  quadric tests", "", "solve: plane cells", "solve: quadric cells", "solve: two-plane cells", "solve in shadow walks", "solve in nearest-hit walks",
            "solve with own / cached transform", "solve with conic fix", "", "cells loaded by packet walks", "  culled by their sphere", "shadow packet walks", "nearest-hit packet walks",
            "trnode cells", "bounding-volume cells", "", "", "shade() calls", "light rounds", "solves without any accepted hit", "  of them planes", "  of them without a candidate root", "box cull tests" };
        for (int i = 0; i < 30; i++) if (nm[i][0]) fprintf(stderr, "QR_PROF %-36s %llu\n", nm[i], pf[i]);
        fprintf(stderr, "QR_PROF algorithmic fp32 operations executed (SURVEY 8(d) weights, per lane) %llu\n", pf[48]);
        fprintf(stderr, "QR_PROF fp32 operations of the implementation's own culls (sphere / box tests, walk set-up) %llu\n", pf[49]);
        fprintf(stderr, "QR_PROF frames pushed by level 0..7+ (lanes), both children:");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", pf[32 + i]);
        fprintf(stderr, "\nQR_PROF frames pushed by level 0..7+ (lanes), one child:    ");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", pf[40 + i]);
        fprintf(stderr, "\n");
        unsigned long long z[64] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(qr_prof), z, sizeof(z)));
    }
#endif
#ifdef QR_STATS
    {
        unsigned long long st[32];
        HIP_TRY(hipMemcpy(st, s->d_counters + 4, sizeof(st), hipMemcpyDeviceToHost));
        if (st[27]) fprintf(stderr, "QR_GUARD %llu bad cell offsets; first: tag %llu offset 0x%llx context 0x%llx\n", st[27], st[24], st[25], st[26]);
        fprintf(stderr, "QR_STATS per-lane walks %llu: steps %llu (%.1f per walk, %.1f lanes stepping), solve rounds %llu (%.1f lanes solving)\n",
                st[16], st[17], st[16] ? (double)st[17] / st[16] : 0.0, st[17] ? (double)st[18] / st[17] : 0.0,
                st[19], st[19] ? (double)st[20] / st[19] : 0.0);
        fprintf(stderr, "QR_STATS per-lane SHADOW walks %llu: steps %llu (%.1f per walk, %.1f lanes stepping, %.1f lanes at start), solve rounds %llu; non-shadow lanes at start %.1f\n",
                st[9], st[10], st[9] ? (double)st[10] / st[9] : 0.0, st[10] ? (double)st[11] / st[10] : 0.0,
                st[9] ? (double)st[15] / st[9] : 0.0, st[22], st[16] ? (double)st[21] / st[16] : 0.0);
        fprintf(stderr, "QR_STATS ranges handed over %llu\n", st[23]);
        fprintf(stderr, "QR_STATS outer rounds %llu: %.1f lanes tracing, %.1f lanes not finished\n", st[28], st[28] ? (double)st[29] / st[28] : 0.0, st[28] ? (double)st[30] / st[28] : 0.0);
        const char *nm[3] = { "shadow", "primary", "secondary" };
        for (int k = 0; k < 3; k++)
            fprintf(stderr, "QR_STATS %s: walks %llu elem-iterations %llu (%.1f per walk, %.0f%% culled) active lanes per iteration %.1f\n", nm[k],
                    st[3 * k], st[3 * k + 1], st[3 * k] ? (double)st[3 * k + 1] / st[3 * k] : 0.0,
                    st[3 * k + 1] ? 100.0 * st[12 + k] / st[3 * k + 1] : 0.0,
                    st[3 * k + 1] ? (double)st[3 * k + 2] / st[3 * k + 1] : 0.0);
    }
#endif
    return QR_OK;
}

extern "C" int qr_render_timed(qr_device_scene *s, void *frame_dev, void *stream,
                               int iters, float *avg_ms, float *min_ms)
{
    if (s == nullptr || frame_dev == nullptr || iters <= 0) return qr_fail(QR_ERR_ARG, "bad argument");
    { const int rc = pt_rows_ok(s); if (rc != QR_OK) return rc; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(s->device));
    double sum = 0.0; float mn = 1e30f;
    if (s->ev0 == nullptr) { HIP_TRY(hipEventCreate(&s->ev0)); HIP_TRY(hipEventCreate(&s->ev1)); }
    for (int i = 0; i < iters; i++)
    {
#ifdef QR_WAVETIME
        HIP_TRY(hipMemsetAsync(s->d_counters + 64, 0, (size_t)s->lp.n_blocks * QR_WT_SLOTS * sizeof(unsigned long long), st));
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
        std::vector<unsigned long long> w((size_t)s->lp.n_blocks * QR_WT_SLOTS);
        HIP_TRY(hipMemcpy(w.data(), s->d_counters + 64, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        FILE *f = fopen(path, "wb");
        if (f) { fwrite(w.data(), sizeof(unsigned long long), w.size(), f); fclose(f); }
    }
#endif
    return QR_OK;
}

#ifdef QR_WAVETIME
/* QR_WAVETIME builds only (tools/gpu_timeline.py): the stamps the waves of this scene's LAST launch wrote, QR_WT_SLOTS words per
 * schedule entry; not part of include/qrhip.h -- the product library does not export it */
extern "C" int qr_wavetime_read(qr_device_scene *s, unsigned long long *out, uint64_t n_words, int clear)
{
    if (s == nullptr || out == nullptr) return qr_fail(QR_ERR_ARG, "bad argument");
    const uint64_t have = (uint64_t)s->lp.n_blocks * QR_WT_SLOTS;
    if (n_words > have) n_words = have;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(out, s->d_counters + 64, n_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (clear) HIP_TRY(hipMemset(s->d_counters + 64, 0, have * sizeof(unsigned long long)));
    return (int)QR_WT_SLOTS;
}
#endif

/* per-thread device frame + pinned staging buffer, kept between calls: the drop-in path renders a frame
 * per call, and hipMalloc/hipFree plus a pageable 8 MB copy cost more than the kernel itself */
#define QR_COPY_CHUNKS 4
#define QR_MAX_DEVICES 8

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

/*
 * QR_DEVICES=0,1,...: the frame of a host-frame call (qr_render0, qr_render_host) is cut into blocks of rows, one run of
 * blocks per entry of the list; every entry renders its rows on its device and copies them into the host frame itself
 * (tracer.cpp:1144-1145 / engine.cpp:3465-3478: the engine's threads own rows index, index + thnum, ... of one frame and write
 * them in place -- here a device owns a band).  The same ordinal may be listed more than once (separate buffers and
 * streams on that device: how a one-GPU box tests the path).  Without QR_DEVICES: the one device of QR_DEVICE (default 0).
 */
static int device_list(std::vector<int> &v)
{
    v.clear();
    if (const char *e = getenv("QR_DEVICES"))
    {
        const char *p = e;
        while (*p)
        {
            while (*p == ',' || *p == ' ') p++;
            if (!*p) break;
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p) return qr_fail(QR_ERR_ARG, std::string("QR_DEVICES: not a list of device ordinals: ") + e);
            /* without any device the call fails as QR_ERR_DEVICE further down (pick_device): no CPU fallback */
            if (d < 0 || (qr_device_count() > 0 && d >= qr_device_count())) return qr_fail(QR_ERR_ARG, std::string("QR_DEVICES: ordinal out of range in ") + e);
            if (v.size() >= QR_MAX_DEVICES) return qr_fail(QR_ERR_ARG, std::string("QR_DEVICES: more than ") + std::to_string(QR_MAX_DEVICES) + " entries");
            v.push_back((int)d);
            p = end;
        }
    }
    if (v.empty()) { int d = 0; if (const char *e = getenv("QR_DEVICE")) d = atoi(e); v.push_back(d); }
    return QR_OK;
}

/*
 * Caller frames registered for direct copies (qr_frame_register): rendered rows go from device memory straight into
 * such a frame by DMA -- no page-locked staging frame, no host copy.  The registration is the CALLER's statement that the
 * range stays mapped until qr_frame_unregister: the library never pins a pointer it merely saw in a call (the engine may
 * free its frame between two calls, engine.cpp:3317-3323 / 2814-2850).
 */
struct PinnedRange { uintptr_t base; size_t bytes; };
static std::vector<PinnedRange> g_pins;
static std::mutex g_pins_lock;

extern "C" int qr_frame_register(void *frame, uint64_t bytes)
{
    if (frame == nullptr || bytes == 0) return qr_fail(QR_ERR_ARG, "null frame");
    std::vector<int> dl;
    int rc = device_list(dl);
    if (rc != QR_OK) return rc;
    rc = pick_device(dl[0]);
    if (rc != QR_OK) return rc;
    std::lock_guard<std::mutex> lk(g_pins_lock);
    for (const PinnedRange &r : g_pins)
        if (r.base == (uintptr_t)frame) return r.bytes == bytes ? QR_OK : qr_fail(QR_ERR_ARG, "frame is registered with another size");
    HIP_TRY(hipHostRegister(frame, (size_t)bytes, hipHostRegisterPortable));
    g_pins.push_back({ (uintptr_t)frame, (size_t)bytes });
    return QR_OK;
}

extern "C" int qr_frame_unregister(void *frame)
{
    std::lock_guard<std::mutex> lk(g_pins_lock);
    for (size_t i = 0; i < g_pins.size(); i++)
        if (g_pins[i].base == (uintptr_t)frame)
        {
            g_pins.erase(g_pins.begin() + (ptrdiff_t)i);
            HIP_TRY(hipHostUnregister(frame));
            return QR_OK;
        }
    return qr_fail(QR_ERR_ARG, "frame was not registered");
}

static bool frame_is_pinned(const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pins_lock);
    for (const PinnedRange &r : g_pins)
        if ((uintptr_t)p >= r.base && (uintptr_t)p + bytes <= r.base + r.bytes) return true;
    return false;
}

/* rows [y0, y1) of a device frame (compact, w pixels a row) that belong to index / thnum, into a registered host frame */
static hipError_t copy_rows_direct(uint32_t *frame_host, int row_pixels, const uint32_t *d_frame, int w, int y0, int y1,
                                   int index, int thnum, hipStream_t st)
{
    if (thnum <= 1)
    {
        if (y1 <= y0) return hipSuccess;
        if (row_pixels == w) return hipMemcpyAsync(frame_host + (size_t)y0 * w, d_frame + (size_t)y0 * w, (size_t)(y1 - y0) * w * 4, hipMemcpyDeviceToHost, st);
        return hipMemcpy2DAsync(frame_host + (size_t)y0 * row_pixels, (size_t)row_pixels * 4, d_frame + (size_t)y0 * w, (size_t)w * 4,
                                (size_t)w * 4, (size_t)(y1 - y0), hipMemcpyDeviceToHost, st);
    }
    const int first = y0 + ((index - y0 % thnum) % thnum + thnum) % thnum;
    if (first >= y1) return hipSuccess;
    const int rows = (y1 - first + thnum - 1) / thnum;
    return hipMemcpy2DAsync(frame_host + (size_t)first * row_pixels, (size_t)thnum * row_pixels * 4, d_frame + (size_t)first * w, (size_t)thnum * w * 4,
                            (size_t)w * 4, (size_t)rows, hipMemcpyDeviceToHost, st);
}

/* rows [y0, y1) of a compact staging frame that belong to index / thnum, into the caller's frame (any stride, also negative:
 * bottom-up frames, engine.cpp:2814-2850) */
static void copy_rows_host(uint32_t *frame_host, int row_pixels, const uint32_t *h_frame, int w, int y0, int y1, int index, int thnum)
{
    if (thnum <= 1 && row_pixels == w) { copy_streaming(frame_host + (size_t)y0 * w, h_frame + (size_t)y0 * w, (size_t)(y1 - y0) * w * 4); return; }
    for (int y = y0; y < y1; y++)
    {
        if (thnum > 1 && (y % thnum) != index) continue;
        memcpy(frame_host + (ptrdiff_t)y * row_pixels, h_frame + (size_t)y * w, (size_t)w * 4);
    }
}

/* bytes of the caller's frame a call may touch (positive stride), 0: not addressable as one range */
static size_t frame_span(int w, int h, int row_pixels)
{
    if (row_pixels < w || h <= 0) return 0;
    return ((size_t)(h - 1) * (size_t)row_pixels + (size_t)w) * 4;
}

/* ---- qr_render_host: uploaded scene -> host frame ---- */

/* what one entry of QR_DEVICES holds for an uploaded scene (several entries only) */
struct HostReplica
{
    int device = -1;
    void *d_blob = nullptr; bool own_blob = false;
    unsigned long long *d_counters = nullptr;
    void *d_frame = nullptr; size_t d_cap = 0;
    uint32_t *d_order = nullptr; int32_t n_order = 0; int y0 = 0, y1 = 0;
    hipStream_t st = nullptr; hipEvent_t ev = nullptr;
};

struct HostPathCache
{
    int device = -1;
    void *d_frame = nullptr; size_t d_cap = 0;
    uint32_t *h_frame = nullptr; size_t h_cap = 0;
    hipEvent_t ev[QR_COPY_CHUNKS] = {};
    /* QR_DEVICES with several entries: replicas of the scene this thread rendered last */
    uint64_t rep_serial = 0;      /* qr_device_scene::serial the replicas were built for (0: none) */
    std::vector<int> rep_devices;
    std::vector<HostReplica> rep;
    ~HostPathCache()
    {
        /* the HIP runtime may already be shut down when thread-locals are destroyed at exit: leak */
    }
};
static thread_local HostPathCache g_hpc;

static void replicas_release(HostPathCache &c)
{
    for (HostReplica &r : c.rep)
    {
        if (r.device < 0 || hipSetDevice(r.device) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        if (r.own_blob) (void)hipFree(r.d_blob);
        (void)hipFree(r.d_counters); (void)hipFree(r.d_frame); (void)hipFree(r.d_order);
        if (r.st) (void)hipStreamDestroy(r.st);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
    c.rep.clear(); c.rep_serial = 0; c.rep_devices.clear();
}

/* the kernel instance an uploaded scene renders with, for an explicit schedule / image / counter block */
static hipError_t launch_instance(const qr_device_scene *s, const LaunchP &lp, uint32_t *frame_dev, unsigned long long *counters, hipStream_t st)
{
    if (lp.n_blocks <= 0) return hipSuccess;
    const dim3 grid((unsigned)((lp.n_blocks + (QR_BLOCK / 64) - 1) / (QR_BLOCK / 64)), 1, 1);
    if (s->divk) hipLaunchKernelGGL((qr_render_kernel<false, QR_DIVK_WAVES, true>), grid, dim3(QR_BLOCK), 0, st, lp, frame_dev, (int32_t *)nullptr, counters);
    else         hipLaunchKernelGGL((qr_render_kernel<false, 4, false>), grid, dim3(QR_BLOCK), 0, st, lp, frame_dev, (int32_t *)nullptr, counters);
    return hipGetLastError();
}

/* (re)build the replicas of `s` for the device list: image copies (peer copy from the scene's device), the schedule entries
 * of every band, a frame buffer, a stream */
static int replicas_prepare(HostPathCache &c, qr_device_scene *s, const std::vector<int> &devs)
{
    /* keyed on the upload's serial: the address of a destroyed scene and of its image are reused by the next upload of the
     * same size (an animation re-uploads every frame), and replicas of the old image must not render for the new one */
    if (c.rep_serial == s->serial && c.rep_devices == devs) return QR_OK;
    replicas_release(c);
    const int n = (int)devs.size(), H = s->fr.frm_h, W = s->fr.frm_w;
    const int fh = s->fr.fsaa == 0 ? 8 : 4;
    c.rep.resize((size_t)n);
    for (int j = 0; j < n; j++)
    {
        HostReplica &r = c.rep[(size_t)j];
        int rc = pick_device(devs[(size_t)j]);
        if (rc != QR_OK) { replicas_release(c); return rc; }
        r.device = devs[(size_t)j];
        /* bands of whole 8-row groups */
        const int groups = (H + 7) / 8;
        r.y0 = (int)((long long)groups * j / n) * 8; r.y1 = j == n - 1 ? H : (int)((long long)groups * (j + 1) / n) * 8;
        if (r.y1 > H) r.y1 = H;
        std::vector<uint32_t> keep;
        for (size_t i = 0; i + 1 < s->h_order.size(); i += 2)
        {
            const int fy = (int)((s->h_order[i] >> 14) & 0x3FFFu) * fh;
            if (fy + fh > r.y0 && fy < r.y1) { keep.push_back(s->h_order[i]); keep.push_back(s->h_order[i + 1]); }
        }
        r.n_order = (int32_t)(keep.size() / 2);
        hipError_t e = hipSuccess;
        if (r.device == s->device) r.d_blob = s->d_blob;
        else
        {
            e = hipMalloc(&r.d_blob, s->blob_bytes);
            if (e == hipSuccess) { r.own_blob = true; e = hipMemcpyPeer(r.d_blob, r.device, s->d_blob, s->device, s->blob_bytes); }
        }
        if (e == hipSuccess) e = hipMalloc((void **)&r.d_counters, 64 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(r.d_counters, 0, 64 * sizeof(unsigned long long));
        if (e == hipSuccess) { r.d_cap = (size_t)W * H * 4; e = hipMalloc(&r.d_frame, r.d_cap); }
        if (e == hipSuccess && r.n_order > 0) e = hipMalloc((void **)&r.d_order, keep.size() * 4);
        if (e == hipSuccess && r.n_order > 0) e = hipMemcpy(r.d_order, keep.data(), keep.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.ev, hipEventDisableTiming);
        if (e != hipSuccess) { replicas_release(c); return qr_fail(QR_ERR_DEVICE, std::string("QR_DEVICES replica: ") + hipGetErrorString(e)); }
    }
    c.rep_serial = s->serial; c.rep_devices = devs;
    return QR_OK;
}

extern "C" int qr_render_host(qr_device_scene *s, uint32_t *frame_host, int row_pixels)
{
    if (s == nullptr || frame_host == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    const int w = s->fr.frm_w, h = s->fr.frm_h;
    const size_t bytes = (size_t)w * h * 4;
    { const int rc = pt_rows_ok(s); if (rc != QR_OK) return rc; }
    HIP_TRY(hipSetDevice(s->device));
    HostPathCache &c = g_hpc;
    const bool whole = s->lp.row_begin == 0 && s->lp.row_end == h && s->lp.group_first == 0 && s->lp.group_stride == 1 && s->lp.thnum <= 1;
    const size_t span = frame_span(w, h, row_pixels);
    const bool pinned = span != 0 && frame_is_pinned(frame_host, span);
    if (c.h_cap < bytes && !pinned)
    {
        if (c.h_frame) (void)hipHostFree(c.h_frame);
        c.h_frame = nullptr; c.h_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c.h_frame, bytes, hipHostMallocPortable));
        c.h_cap = bytes;
    }

    /* ---- several devices: one band of rows each ---- */
    std::vector<int> devs;
    if (getenv("QR_DEVICES")) { const int rc = device_list(devs); if (rc != QR_OK) return rc; }
    if (devs.size() > 1 && whole && !s->pt_on)
    {
        int rc = replicas_prepare(c, s, devs);
        if (rc != QR_OK) return rc;
        hipError_t e = hipSuccess;
        for (HostReplica &r : c.rep)
        {
            if (e != hipSuccess) break;
            e = hipSetDevice(r.device);
            if (e != hipSuccess) break;
            LaunchP lp = s->lp;
            lp.B = (const char *)r.d_blob; lp.order = r.d_order; lp.n_blocks = r.n_order;
            lp.row_begin = r.y0; lp.row_end = r.y1; lp.stats = r.d_counters + 4;
            e = launch_instance(s, lp, (uint32_t *)r.d_frame, r.d_counters, r.st);
            if (e != hipSuccess) break;
            if (pinned) e = copy_rows_direct(frame_host, row_pixels, (const uint32_t *)r.d_frame, w, r.y0, r.y1, 0, 1, r.st);
            else if (r.y1 > r.y0) e = hipMemcpyAsync(c.h_frame + (size_t)r.y0 * w, (const uint32_t *)r.d_frame + (size_t)r.y0 * w, (size_t)(r.y1 - r.y0) * w * 4, hipMemcpyDeviceToHost, r.st);
            if (e == hipSuccess) e = hipEventRecord(r.ev, r.st);
        }
        for (HostReplica &r : c.rep)
        {
            if (e != hipSuccess) break;
            e = hipEventSynchronize(r.ev);
            if (e == hipSuccess && !pinned) copy_rows_host(frame_host, row_pixels, c.h_frame, w, r.y0, r.y1, 0, 1);
        }
        (void)hipSetDevice(s->device);
        if (e != hipSuccess) return qr_fail(QR_ERR_DEVICE, std::string("render (QR_DEVICES): ") + hipGetErrorString(e));
        return QR_OK;
    }

    if (c.device != s->device || c.d_cap < bytes)
    {
        if (c.d_frame) { (void)hipSetDevice(c.device); (void)hipFree(c.d_frame); (void)hipSetDevice(s->device); }
        c.d_frame = nullptr; c.d_cap = 0; c.device = s->device;
        HIP_TRY(hipMalloc(&c.d_frame, bytes));
        c.d_cap = bytes;
    }
    hipError_t e = launch<false>(s, c.d_frame, nullptr, nullptr);
    if (e == hipSuccess && pinned)
    {
        /* registered frame: the rows this call owns by DMA, straight into it */
        if (s->lp.group_stride == 1)
            e = copy_rows_direct(frame_host, row_pixels, (const uint32_t *)c.d_frame, w, s->lp.row_begin, s->lp.row_end, s->lp.index, s->lp.thnum, nullptr);
        else
            for (int g = s->lp.group_first; g * 8 < h && e == hipSuccess; g += s->lp.group_stride)
                e = copy_rows_direct(frame_host, row_pixels, (const uint32_t *)c.d_frame, w, g * 8, g * 8 + 8 < h ? g * 8 + 8 : h, s->lp.index, s->lp.thnum, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return qr_fail(QR_ERR_DEVICE, std::string("render: ") + hipGetErrorString(e));
        return QR_OK;
    }
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
    for (int y = s->lp.row_begin; y < s->lp.row_end; y++)
    {
        if ((y / 8 - s->lp.group_first) % s->lp.group_stride != 0 || y / 8 < s->lp.group_first) continue;
        if (s->lp.thnum > 1 && (y % s->lp.thnum) != s->lp.index) continue;
        memcpy(frame_host + (ptrdiff_t)y * row_pixels, c.h_frame + (size_t)y * w, (size_t)w * 4);
    }
    return QR_OK;
}

/*
 * The reference entry point.  One call = flatten + compile + upload + launch + copy back; nothing of the
 * caller's memory is retained (the engine releases its pools every frame, engine.cpp:3317-3323).
 *
 * What IS kept, per calling thread, is our own: per entry of QR_DEVICES (one entry without it) a device arena for the
 * scene image and the frame (no hipMalloc / hipFree per call) and streams; page-locked staging buffers and the vectors
 * of the host passes.  The frame is rendered in horizontal blocks (the schedule is grouped by block), QR_DROPIN_BLOCKS of
 * them on one device, a run of blocks per device on several: block k is copied back over PCIe on its device's copy
 * stream while block k + 1 renders, and the host moves finished blocks into the caller's frame (non-temporal stores)
 * while the next copy is in flight -- or, when the caller has registered its frame (qr_frame_register), the copy engines
 * write the blocks straight into it.  When the flattened scene is byte-identical to the one this thread uploaded last (a
 * paused animation), validation, compilation and upload are skipped.
 */
#ifndef QR_DROPIN_BLOCKS
#define QR_DROPIN_BLOCKS 4
#ifndef QR_DROPIN_STREAMS
#define QR_DROPIN_STREAMS 3     /* measured through the engine, demo1 / demo1 + swarm, ms per frame: 1: 0.83 / 14.1, 2: 0.75 / 12.3, 3: 0.76 / 11.3, 4: 0.83 / 12.1 */
#endif
#endif

/* what one entry of the device list holds on its device */
struct DropSlot
{
    int device = -1;
    void *d_blob = nullptr; size_t d_cap = 0;
    void *d_frame = nullptr; size_t df_cap = 0;
    unsigned long long *d_counters = nullptr; size_t n_counters = 0;
    hipStream_t sk = nullptr, sc = nullptr;
    hipStream_t sx[QR_DROPIN_BLOCKS] = {};     /* one launch stream per row block (sx[0] == sk): a block's tail overlaps the next block's bulk */
    hipEvent_t ev_up = nullptr;                 /* uploads of this call are on the device */
    hipEvent_t ev_k[QR_DROPIN_BLOCKS] = {}, ev_c[QR_DROPIN_BLOCKS] = {};
    bool resident = false;              /* the thread's current program is the image in d_blob */
    uint32_t *d_pt = nullptr; size_t pt_cap = 0;   /* path-tracer mode: device copies of the engine's seed plane and three colour planes */
};

struct DropIn
{
    DropSlot slot[QR_MAX_DEVICES];
    int n_slots = 0;
    uint8_t *h_stage = nullptr; size_t h_cap = 0;      /* page-locked, portable: every device uploads from it */
    uint32_t *h_frame = nullptr; size_t hf_cap = 0;
    std::vector<uint8_t> blob, last_blob;
    QrProgram prog;
    bool compiled = false;              /* prog is the program of last_blob ... */
    int compiled_blocks = 0;            /* ... with its schedule grouped for this many row blocks */
};
static thread_local DropIn g_drop;

/* everything a slot holds on its device; the caller has made that device current */
static void slot_release(DropSlot &c)
{
    (void)hipDeviceSynchronize();
    for (int k = 1; k < QR_DROPIN_BLOCKS; k++) if (c.sx[k]) (void)hipStreamDestroy(c.sx[k]);
    if (c.sk) (void)hipStreamDestroy(c.sk);
    if (c.sc) (void)hipStreamDestroy(c.sc);
    if (c.ev_up) (void)hipEventDestroy(c.ev_up);
    for (int k = 0; k < QR_DROPIN_BLOCKS; k++) { if (c.ev_k[k]) (void)hipEventDestroy(c.ev_k[k]); if (c.ev_c[k]) (void)hipEventDestroy(c.ev_c[k]); }
    (void)hipFree(c.d_blob); (void)hipFree(c.d_frame); (void)hipFree(c.d_counters); (void)hipFree(c.d_pt);
    c = DropSlot();
}

/* streams, events: created together; on any failure nothing is kept (the slot's device stays -1, the next call starts over) */
static int slot_create(DropSlot &c)
{
    HIP_TRY(hipStreamCreateWithFlags(&c.sk, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c.sc, hipStreamNonBlocking));
    c.sx[0] = c.sk;
    for (int k = 1; k < QR_DROPIN_BLOCKS; k++) HIP_TRY(hipStreamCreateWithFlags(&c.sx[k], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c.ev_up, hipEventDisableTiming));
    for (int k = 0; k < QR_DROPIN_BLOCKS; k++)
    {
        HIP_TRY(hipEventCreateWithFlags(&c.ev_k[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c.ev_c[k], hipEventDisableTiming));
    }
    return QR_OK;
}

static int slot_prepare(DropSlot &c, int dev, size_t image_bytes, size_t frame_bytes, size_t n_sched)
{
    if (c.device != dev)
    {
        /* first call of this thread, or another device: what the previous device held is released there first;
         * c.device is only set once every stream and event exists, so a failure leaves no half-built state behind */
        if (c.device >= 0) { if (hipSetDevice(c.device) == hipSuccess) slot_release(c); c.device = -1; }
        int rc = pick_device(dev);
        if (rc != QR_OK) return rc;
        rc = slot_create(c);
        if (rc != QR_OK) { slot_release(c); return rc; }
        c.device = dev;
    }
    else HIP_TRY(hipSetDevice(dev));
    {
        /* counter block as qr_scene_upload_ex sizes it: 64 words (ray counts, QR_STATS slots 4..34), and per wave of the
         * schedule QR_WT_SLOTS more in QR_WAVETIME builds */
#ifdef QR_WAVETIME
        const size_t need = 64 + QR_WT_SLOTS * n_sched;
#else
        const size_t need = 64; (void)n_sched;
#endif
        if (c.n_counters < need)
        {
            if (c.d_counters) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c.d_counters); }
            c.d_counters = nullptr; c.n_counters = 0;
            HIP_TRY(hipMalloc((void **)&c.d_counters, need * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(c.d_counters, 0, need * sizeof(unsigned long long)));
            c.n_counters = need;
        }
    }
    if (c.d_cap < image_bytes)
    {
        if (c.d_blob) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c.d_blob); }
        c.d_blob = nullptr; c.d_cap = 0; c.resident = false;
        const size_t cap = image_bytes + image_bytes / 2 + (1u << 20);
        HIP_TRY(hipMalloc(&c.d_blob, cap));
        c.d_cap = cap;
    }
    if (c.df_cap < frame_bytes)
    {
        if (c.d_frame) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c.d_frame); }
        c.d_frame = nullptr; c.df_cap = 0;
        HIP_TRY(hipMalloc(&c.d_frame, frame_bytes));
        c.df_cap = frame_bytes;
    }
    return QR_OK;
}

/* the thread's page-locked staging buffers (some device is current) */
static int dropin_host_buffers(DropIn &c, size_t image_bytes, size_t frame_bytes)
{
    if (c.h_cap < image_bytes)
    {
        if (c.h_stage) (void)hipHostFree(c.h_stage);
        c.h_stage = nullptr; c.h_cap = 0;
        for (int j = 0; j < QR_MAX_DEVICES; j++) c.slot[j].resident = false;       /* the image every upload reads is gone */
        const size_t cap = image_bytes + image_bytes / 2 + (1u << 20);
        HIP_TRY(hipHostMalloc((void **)&c.h_stage, cap, hipHostMallocPortable));
        c.h_cap = cap;
    }
    if (c.hf_cap < frame_bytes)
    {
        if (c.h_frame) (void)hipHostFree(c.h_frame);
        c.h_frame = nullptr; c.hf_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c.h_frame, frame_bytes, hipHostMallocPortable));
        c.hf_cap = frame_bytes;
    }
    return QR_OK;
}

/*
 * Path-tracer mode through the drop-in entry point.  The engine keeps one LCG state per pixel sample (inf_PSEED) and three
 * colour planes with the running mean (inf_PTR_R/G/B, engine.cpp:2875-2893), all frm_row << fsaa words per row, and a
 * sample counter in every thread's rt_SIMD_INFOX which render0 itself advances (tracer.cpp:1112-1136).  A call owns rows
 * index, index + thnum, ...: those rows of the four planes travel to the device before the launches and back after them.
 */
struct DropInPt
{
    uint8_t *host[4] = { nullptr, nullptr, nullptr, nullptr };     /* seeds, r, g, b on the host */
    size_t row_bytes = 0, plane_words = 0;
    int first = 0, step = 1, rows = 0;
};

static int dropin_pt_begin(DropSlot &c, const void *s_inf, const qr_abi_desc *abi, const qr_frame &fr, DropInPt &io, PtParams &pt)
{
    uint8_t *inf = (uint8_t *)(uintptr_t)s_inf;         /* the counter below is the one field of s_inf render0 writes */
    const size_t ps = abi->pointer_bits / 8, Q = abi->quads, P = abi->pointer_bits / 32;
    const size_t ib = Q * 0x100;
    static const int slot[4] = { 3, 16, 17, 18 };       /* inf_PSEED, inf_PTR_R, inf_PTR_G, inf_PTR_B (tracer.h:163, 205-211) */
    for (int k = 0; k < 4; k++)
    {
        uint64_t p = 0;
        if (ps == 8) memcpy(&p, inf + ib + (size_t)slot[k] * ps, 8); else { uint32_t a; memcpy(&a, inf + ib + (size_t)slot[k] * ps, 4); p = a; }
        if (p == 0) return qr_fail(QR_ERR_ARG, "path-tracer mode without seed / colour planes in s_inf");
        io.host[k] = (uint8_t *)(uintptr_t)p;
    }
    /* sample count + 1, weights of the running mean: inf_PTS_C / _O / _U, lane-broadcast (tracer.h:251-257) */
    float *pts_c = (float *)(inf + Q * 0x130 + 0x100 * P), *pts_o = (float *)(inf + Q * 0x140 + 0x100 * P),
          *pts_u = (float *)(inf + Q * 0x150 + 0x100 * P);
    const float cnt = pts_c[0] + 1.0f, o = 1.0f / cnt, u = 1.0f - o;
    for (size_t l = 0; l < Q * 4; l++) { pts_c[l] = cnt; pts_o[l] = o; pts_u[l] = u; }

    const size_t ns = (size_t)1 << fr.fsaa;
    io.row_bytes = (size_t)fr.frm_row * ns * 4;
    io.plane_words = (size_t)fr.frm_row * fr.frm_h * ns;
    io.step = fr.thnum > 0 ? fr.thnum : 1;
    io.first = fr.index;
    io.rows = fr.frm_h > io.first ? (fr.frm_h - io.first + io.step - 1) / io.step : 0;
    if (c.pt_cap < 4 * io.plane_words)
    {
        if (c.d_pt) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c.d_pt); }
        c.d_pt = nullptr; c.pt_cap = 0;
        HIP_TRY(hipMalloc((void **)&c.d_pt, 4 * io.plane_words * 4));
        c.pt_cap = 4 * io.plane_words;
    }
    for (int k = 0; k < 4 && io.rows > 0; k++)
        HIP_TRY(hipMemcpy2DAsync((uint8_t *)(c.d_pt + (size_t)k * io.plane_words) + (size_t)io.first * io.row_bytes, io.step * io.row_bytes,
                                 io.host[k] + (size_t)io.first * io.row_bytes, io.step * io.row_bytes,
                                 io.row_bytes, (size_t)io.rows, hipMemcpyHostToDevice, c.sk));
    pt.seeds = c.d_pt;
    pt.acc_r = (float *)(c.d_pt + io.plane_words); pt.acc_g = (float *)(c.d_pt + 2 * io.plane_words); pt.acc_b = (float *)(c.d_pt + 3 * io.plane_words);
    pt.pts_o = o; pt.pts_u = u;
    pt.eager = 1;               /* the reference's shading order: its random streams, its frames */
    pt.pad = 0;
    return QR_OK;
}

static hipError_t dropin_pt_end(DropSlot &c, const DropInPt &io)
{
    hipError_t e = hipSuccess;
    for (int k = 0; k < 4 && io.rows > 0 && e == hipSuccess; k++)
        e = hipMemcpy2DAsync(io.host[k] + (size_t)io.first * io.row_bytes, io.step * io.row_bytes,
                             (const uint8_t *)(c.d_pt + (size_t)k * io.plane_words) + (size_t)io.first * io.row_bytes, io.step * io.row_bytes,
                             io.row_bytes, (size_t)io.rows, hipMemcpyDeviceToHost, c.sk);
    if (e == hipSuccess) e = hipStreamSynchronize(c.sk);
    return e;
}

extern "C" int qr_render0(const void *s_inf, const qr_abi_desc *abi)
{
    const bool verbose = getenv("QR_VERBOSE") != nullptr;
    const double t0 = now_ms();
    DropIn &c = g_drop;
    std::string err;
    int rc = qr_flatten_impl(s_inf, abi, c.blob, err);
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
    uint32_t *frame_host = (uint32_t *)(uintptr_t)p_frame;
    const int row_pixels = (int)row;

    /* the devices of this call, and with them the number of row blocks of the schedule: QR_DROPIN_BLOCKS on one device,
     * a run of blocks (at least one) for each of several.  A path-traced frame stays on the first device: its sample planes
     * live there */
    std::vector<int> devs;
    rc = device_list(devs);
    if (rc != QR_OK) return rc;
    qr_scene_view v;
    rc = qr_scene_view_init(&v, c.blob.data(), c.blob.size());
    if (rc != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot (qr_scene_view_init " + std::to_string(rc) + ")");
    if (v.frame->pt_on) devs.resize(1);
    const int n_dev = (int)devs.size();
    const int want_blocks = n_dev * (n_dev >= QR_DROPIN_BLOCKS ? 1 : QR_DROPIN_BLOCKS / n_dev);

    /* host passes (skipped when nothing changed since this thread's last call) */
    const bool same = c.compiled && c.compiled_blocks == want_blocks && c.blob.size() == c.last_blob.size()
                   && memcmp(c.blob.data(), c.last_blob.data(), c.blob.size()) == 0;
    if (!same)
    {
        c.compiled = false;
        for (int j = 0; j < QR_MAX_DEVICES; j++) c.slot[j].resident = false;
        rc = qr_snapshot_validate(v, err);
        if (rc != QR_OK) return qr_fail(rc, err);
        static thread_local std::vector<BSphere> bsph;
        static thread_local std::vector<qr_elem> E;
        static thread_local std::vector<int32_t> T;
        qr_bound_spheres(v, bsph);
        static const bool rebin = []() { const char *v = getenv("QR_REBIN"); return v && atoi(v) != 0; }();
        if (rebin) E.reserve((size_t)v.hdr->n_elm + (size_t)v.hdr->n_elm / 2 + 65536);      /* room for the tile cells */
        E.assign(v.elm, v.elm + v.hdr->n_elm);
        T.assign(v.tiles, v.tiles + v.hdr->n_tiles);
        qr_frame frm = *v.frame;
        /* QR_REBIN=1: the per-tile lists are rebuilt on the GPU from the camera list instead of taken from the engine (the
         * engine may then run without its own tiling, RT_OPTS_TILING).  The frame is the engine's UNTILED picture: on most
         * scenes that is also its tiled one, but where its screen tiling drops a surface from a tile that the surface does
         * cover the two differ, so this is not the drop-in-exact mode */
        if (rebin)
        {
            rc = pick_device(devs[0]);
            if (rc != QR_OK) return rc;
            rc = rebin_tiles(v, bsph, frm, E, T);
            if (rc != QR_OK) return rc;
        }
        /* the finished image is verified offset by offset (qr_program_verify) on EVERY frame that was compiled: the layout
         * depends on list contents (programs shared by content, clear runs, box cells), so equal totals do not mean an equal
         * structure, and the product kernel checks no offset it loads.  The walk costs ~0.05 ms at 1080p since heads equal to
         * the one just checked are skipped (tools/host_compile_time.py).  QR_VERIFY=0 switches it off (timing experiments) */
        static const bool verify_on = []() { const char *e = getenv("QR_VERIFY"); return e == nullptr || atoi(e) != 0; }();
        rc = qr_program_build(v, E, T, frm, bsph, c.prog, err, want_blocks, false);
        if (rc != QR_OK) return qr_fail(rc, err);
        if (verify_on)
        {
            rc = qr_program_verify(c.prog, err);
            if (rc != QR_OK) return qr_fail(rc, err);
        }
    }
    const double t2 = now_ms();
    const qr_frame &fr = c.prog.frm;
    const int w = fr.frm_w, h = fr.frm_h;
    const size_t frame_bytes = (size_t)w * h * 4;
    const int K = (int)c.prog.block_first.size() - 1;          /* fewer than asked for on a frame of few tile rows */
    const int bps = (K + n_dev - 1) / n_dev;                   /* blocks per slot: block k belongs to slot k / bps */
    if (K < 1 || bps > QR_DROPIN_BLOCKS) return qr_fail(QR_ERR_ARG, "schedule blocks do not match the device list");
    /* slots in use: entry j of the list; slots beyond the list release what they hold */
    for (int j = 0; j < n_dev; j++)
    {
        rc = slot_prepare(c.slot[j], devs[(size_t)j], c.prog.blob.size(), frame_bytes, c.prog.n_sched);
        if (rc != QR_OK) return rc;
    }
    for (int j = n_dev; j < c.n_slots; j++)
        if (c.slot[j].device >= 0) { if (hipSetDevice(c.slot[j].device) == hipSuccess) slot_release(c.slot[j]); c.slot[j] = DropSlot(); }
    c.n_slots = n_dev;
    const size_t span = frame_span(w, h, row_pixels);
    const bool pinned = span != 0 && frame_is_pinned(frame_host, span);
    rc = dropin_host_buffers(c, c.prog.blob.size(), pinned ? 0 : frame_bytes);
    if (rc != QR_OK) return rc;
    if (!same)
    {
        memcpy(c.h_stage, c.prog.blob.data(), c.prog.blob.size());
        c.last_blob.swap(c.blob);
        c.compiled = true; c.compiled_blocks = want_blocks;
    }
    hipError_t e = hipSuccess;
    for (int j = 0; j < n_dev && e == hipSuccess; j++)
    {
        DropSlot &sl = c.slot[j];
        e = hipSetDevice(sl.device);
        if (e == hipSuccess && !sl.resident)
        {
            e = hipMemcpyAsync(sl.d_blob, c.h_stage, c.prog.blob.size(), hipMemcpyHostToDevice, sl.sk);
            sl.resident = e == hipSuccess;
        }
        if (e == hipSuccess) e = hipEventRecord(sl.ev_up, sl.sk);
    }
    if (e != hipSuccess) { for (int j = 0; j < n_dev; j++) c.slot[j].resident = false; return qr_fail(QR_ERR_DEVICE, std::string("qr_render0 upload: ") + hipGetErrorString(e)); }
    const double t3 = now_ms();

    DropInPt ptio; PtParams pt = {};
    if (fr.pt_on)
    {
        HIP_TRY(hipSetDevice(c.slot[0].device));
        rc = dropin_pt_begin(c.slot[0], s_inf, abi, fr, ptio, pt);
        if (rc != QR_OK) return rc;
        /* the seed and colour planes travel on the slot's first stream AFTER the image: the launch streams of the later row
         * blocks wait on ev_up, so it is recorded again behind those copies (round 4: recorded only behind the image, a block
         * on sx[1] / sx[2] could start before its seeds had arrived -- one wrong frame in ~20 runs with four engine threads) */
        HIP_TRY(hipEventRecord(c.slot[0].ev_up, c.slot[0].sk));
    }
    else
    {
        /* FF_ini, tracer.cpp:1128-1132: every ray-traced frame zeroes the sample count in s_inf; rt_Scene::set_pton relies on
         * it to restart the accumulation after it has reset the seed and colour planes (engine.cpp:3729-3745) */
        float *pts_c = (float *)((uint8_t *)(uintptr_t)s_inf + (size_t)abi->quads * 0x130 + 0x100 * (size_t)(abi->pointer_bits / 32));
        for (size_t l = 0; l < (size_t)abi->quads * 4; l++) pts_c[l] = 0.0f;
    }

    LaunchP lp = {};
    lp.depth = fr.depth > QR_MAX_DEPTH ? QR_MAX_DEPTH : fr.depth;
    lp.index = fr.index; lp.thnum = fr.thnum > 0 ? fr.thnum : 1;
    lp.group_first = 0; lp.group_stride = 1;
    /* kernel instance as for uploaded scenes: QR_DIV=0 / 1 forces it (experiments, tests); shadow grids need the per-lane one */
    static const int force_div = []() { const char *dv = getenv("QR_DIV"); return dv ? (atoi(dv) != 0 ? 1 : 0) : -1; }();
    const bool divk = (force_div >= 0 ? force_div != 0 : c.prog.has_long_lists) || c.prog.has_grids;
    /* QR_DROPIN_STREAMS=n: a slot's blocks on n streams in turn (1: one after the other) */
    static const int n_streams = []() { const char *v = getenv("QR_DROPIN_STREAMS"); const int n = v ? atoi(v) : QR_DROPIN_STREAMS;
                                        return n < 1 ? 1 : (n > QR_DROPIN_BLOCKS ? QR_DROPIN_BLOCKS : n); }();
    for (int k = 0; k < K && e == hipSuccess; k++)
    {
        DropSlot &sl = c.slot[k / bps];
        const int kk = k % bps;
        e = hipSetDevice(sl.device);
        if (e != hipSuccess) break;
        hipStream_t sk = sl.sx[kk % n_streams];
        if (sk != sl.sk) e = hipStreamWaitEvent(sk, sl.ev_up, 0);
        if (e != hipSuccess) break;
        const uint32_t e0 = c.prog.block_first[k], e1 = c.prog.block_first[k + 1];
        lp.B = (const char *)sl.d_blob;
        lp.stats = sl.d_counters + 4;
        lp.order = (const uint32_t *)((const char *)sl.d_blob + c.prog.off_order) + 2 * (size_t)e0;
        lp.n_blocks = (int32_t)(e1 - e0);
        lp.row_begin = (int32_t)c.prog.block_row[k]; lp.row_end = (int32_t)c.prog.block_row[k + 1];
        if (lp.n_blocks > 0)
        {
            if (fr.pt_on)
                hipLaunchKernelGGL(qr_render_pt_kernel, dim3((unsigned)lp.n_blocks), dim3(QR_BLOCK), 0, sk,
                                   lp, pt, (uint32_t *)sl.d_frame, sl.d_counters);
            else if (divk)
                hipLaunchKernelGGL((qr_render_kernel<false, QR_DIVK_WAVES, true>), dim3((unsigned)lp.n_blocks), dim3(QR_BLOCK), 0, sk,
                                   lp, (uint32_t *)sl.d_frame, (int32_t *)nullptr, sl.d_counters);
            else
                hipLaunchKernelGGL((qr_render_kernel<false, 4, false>), dim3((unsigned)lp.n_blocks), dim3(QR_BLOCK), 0, sk,
                                   lp, (uint32_t *)sl.d_frame, (int32_t *)nullptr, sl.d_counters);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(sl.ev_k[kk], sk);
        /* copy block k back as soon as it is rendered, on the slot's copy stream: into the caller's registered frame, or
         * into the staging frame */
        if (e == hipSuccess) e = hipStreamWaitEvent(sl.sc, sl.ev_k[kk], 0);
        if (e == hipSuccess)
        {
            if (pinned) e = copy_rows_direct(frame_host, row_pixels, (const uint32_t *)sl.d_frame, w, lp.row_begin, lp.row_end, lp.index, lp.thnum, sl.sc);
            else if (lp.row_end > lp.row_begin)
                e = hipMemcpyAsync(c.h_frame + (size_t)lp.row_begin * w, (const uint32_t *)sl.d_frame + (size_t)lp.row_begin * w,
                                   (size_t)(lp.row_end - lp.row_begin) * w * 4, hipMemcpyDeviceToHost, sl.sc);
        }
        if (e == hipSuccess) e = hipEventRecord(sl.ev_c[kk], sl.sc);
    }
    if (fr.pt_on && e == hipSuccess)
    {
        DropSlot &sl = c.slot[0];
        for (int k = 0; k < K && e == hipSuccess; k++) e = hipStreamWaitEvent(sl.sk, sl.ev_k[k % bps], 0);   /* the planes go back when every block is done */
        if (e == hipSuccess) e = dropin_pt_end(sl, ptio);    /* the planes with this frame's sample go back to the engine */
    }
    const double t4 = now_ms();
    /* the host moves finished blocks into the caller's frame: only the rows this call owns (index / thnum),
     * honouring a negative stride (bottom-up frames, engine.cpp:2814-2850); a registered frame only waits for its copies */
    for (int k = 0; k < K && e == hipSuccess; k++)
    {
        e = hipEventSynchronize(c.slot[k / bps].ev_c[k % bps]);
        if (e != hipSuccess) break;
        if (!pinned) copy_rows_host(frame_host, row_pixels, c.h_frame, w, (int)c.prog.block_row[k], (int)c.prog.block_row[k + 1], lp.index, lp.thnum);
    }
    if (e != hipSuccess)
    {
        for (int j = 0; j < n_dev; j++) c.slot[j].resident = false;
        return qr_fail(QR_ERR_DEVICE, std::string("qr_render0: ") + hipGetErrorString(e));
    }
    if (verbose)
        fprintf(stderr, "qr_render0: flatten %.3f ms (%zu bytes), validate+compile %.3f ms%s, stage+upload %.3f ms (%zu bytes), launches %.3f ms, wait+copy-back %.3f ms%s, %d device slot%s\n",
                t1 - t0, c.last_blob.size(), t2 - t1, same ? " (unchanged scene: skipped)" : "", t3 - t2, c.prog.blob.size(), t4 - t3, now_ms() - t4,
                pinned ? " (registered frame: direct)" : "", n_dev, n_dev == 1 ? "" : "s");
    return QR_OK;
}
