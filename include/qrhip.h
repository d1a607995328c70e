/*
 * qrhip.h - C ABI of the MI355X (gfx950) rendering backend for QuadRay.
 *
 * Drop-in boundary.  The reference selects a rendering backend in
 * rt_Platform::render0 (core/tracer/tracer.cpp:5992-6104) by switching on
 * s_mode and calling `simd_<W>v<V>::render0(rt_SIMD_INFOX *s_inf)`
 * (declared tracer.cpp:5882-5985, defined by each tracer_<W>v<V>.cpp through
 * `#include "tracer.cpp"`, e.g. tracer_128v4.cpp:50-53, body tracer.cpp:1081).
 * `qr_render0` below is what such a namespace symbol forwards to; the ~20 line
 * forwarding TU is shown in INTEGRATION.md (and built by oracle/Makefile as
 * oracle/ref_shim.cpp for the in-container link test).
 *
 * All entry points are plain C: pointers, sizes, ints.  No torch / HIP types.
 * Return value: 0 on success, negative qr_status on failure;
 * qr_last_error() gives a thread-local human readable message.
 */
#ifndef QRHIP_H
#define QRHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qr_status
{
    QR_OK            =  0,
    QR_ERR_ARG       = -1,  /* bad argument / malformed snapshot            */
    QR_ERR_ABI       = -2,  /* unsupported reference build configuration    */
    QR_ERR_UNSUP     = -3,  /* outside what this entry point does (e.g. 8x FSAA, counting renders in path-tracer mode) */
    QR_ERR_DEVICE    = -4,  /* no usable HIP device / HIP runtime error     */
    QR_ERR_IO        = -5,
    QR_ERR_NOMEM     = -6
} qr_status;

/*
 * Build parameters of the reference binary that produced `s_inf`.
 * They fix every structure offset in tracer.h (DP(Q*0x..), P, E macros).
 *   quads         RT_SIMD_QUADS / Q   (rtbase.h:275-277; 8 in the stock x64 build)
 *   pointer_bits  RT_POINTER          (32/64)
 *   address_bits  RT_ADDRESS          (32/64)
 *   element_bits  RT_ELEMENT          (32 only; fp64 builds are rejected)
 *   endian        RT_ENDIAN           (0 little)
 */
typedef struct qr_abi_desc
{
    uint32_t struct_size;   /* = sizeof(qr_abi_desc) */
    uint32_t quads;
    uint32_t pointer_bits;
    uint32_t address_bits;
    uint32_t element_bits;
    uint32_t endian;
    uint32_t reserved[2];
} qr_abi_desc;

/* ------------------------------------------------------------------------ */
/* 1. The reference entry point                                              */
/* ------------------------------------------------------------------------ */

/*
 * Replacement for `simd_<W>v<V>::render0(rt_SIMD_INFOX*)`, tracer.cpp:1081.
 * Reads the structure graph under s_inf (read-only), renders rows
 * index, index+thnum, ... of the frame on the current HIP device and writes
 * 0x00RRGGBB pixels to s_inf->frame (host memory, stride frm_row).
 * Re-entrant; retains no pointer after returning.
 * Path-tracer mode (s_inf->pt_on, tracer.h:214): as render0 does, the call advances the sample counter in s_inf
 * (inf_PTS_C/_O/_U) and adds one sample to the engine's seed and colour planes (inf_PSEED, inf_PTR_R/G/B): its rows of
 * those planes travel to the device and back; the frames are the reference's bit for bit.
 */
int qr_render0(const void *s_inf, const qr_abi_desc *abi);

/*
 * Several devices and caller-owned frames (both optional; they apply to qr_render0 and qr_render_host alike).
 *
 * QR_DEVICES=0,1,... (environment): the frame is cut into bands of tile rows, one band per entry of the list; every entry
 * renders its band on its device and copies it into the host frame itself -- no exchange between devices, the bands meet in
 * the caller's frame as the rows of the engine's worker threads do (tracer.cpp:1144-1145, engine.cpp:3465-3478).  The same
 * ordinal may be listed several times (separate buffers and streams on that device).  Without it: QR_DEVICE (default 0).
 *
 * qr_frame_register: page-locks [frame, frame + bytes) so that rendered rows are written into it by the devices' copy
 * engines directly (no staging frame, no host copy; strides and index / thnum row ownership are honoured).  It is the
 * caller's promise that the range stays mapped until qr_frame_unregister; qr_render0 itself never retains a pointer it
 * was handed (the engine may free its frame between two calls, engine.cpp:3317-3323).  The binding calls it where the
 * reference allocates the frame and the inverse where it frees it (rt_Scene's constructor / destructor, engine.cpp:2829-
 * 2858 / 3815: INTEGRATION.md).  Frames with a negative stride are served through the staging path.
 */
int qr_frame_register(void *frame, uint64_t bytes);
int qr_frame_unregister(void *frame);

/*
 * Same walk as qr_render0 but, instead of rendering, serialises the flattened
 * scene (include/qr_scene.h) to `path`.  Needs no GPU.  This is how snapshots
 * travel from a machine that has the reference engine to one that does not.
 */
int qr_capture_snapshot(const void *s_inf, const qr_abi_desc *abi, const char *path);

/*
 * Which snapshot index the last qr_capture_snapshot on this thread gave to one of the engine's records:
 * kind 0 = rt_SIMD_SURFACE (tracer.h:821) -> qr_surface index, the value hit-id planes carry;
 * kind 1 = rt_SIMD_LIGHT (tracer.h:765) -> qr_light index.  -1 when the record is not part of the snapshot.
 * This is how a host maps the objects of its hierarchy (include/qr_hierarchy.h) to snapshot records.
 */
int qr_capture_index(int kind, const void *record);

/* As above into a malloc'ed buffer the caller releases with qr_free. */
int qr_flatten(const void *s_inf, const qr_abi_desc *abi, void **blob, uint64_t *size);
void qr_free(void *blob);

/* ------------------------------------------------------------------------ */
/* 2. Snapshot-driven rendering (what tests/bench use on the GPU box)        */
/* ------------------------------------------------------------------------ */

typedef struct qr_device_scene qr_device_scene; /* opaque, device resident */

typedef struct qr_scene_info
{
    int32_t frm_w, frm_h, fsaa, depth;
    int32_t n_srf, n_mat, n_lgt, n_elm, n_tiles, n_texels;
    int32_t tile_w, tile_h;
    uint64_t device_bytes;      /* bytes resident in HBM for this scene */
} qr_scene_info;

/* per-kind ray counters (the reference has none; see DESIGN.md "rays") */
typedef struct qr_ray_counts
{
    uint64_t primary;
    uint64_t shadow;
    uint64_t reflect;
    uint64_t refract;
} qr_ray_counts;

/* Upload a snapshot blob to `device` (HIP ordinal). */
int qr_scene_upload(const void *blob, uint64_t size, int device, qr_device_scene **out);

/*
 * As qr_scene_upload with options.  QR_UPLOAD_REBIN_TILES discards the snapshot's per-tile
 * surface lists and rebuilds them on the GPU from the camera list `clist` (replaces the tile
 * binning of rt_Scene::render / render_slice, engine.cpp:1956-2128, 3129-3253): every surface's
 * conservative screen rectangle is matched against every tile by a binning kernel that keeps the
 * camera list's order and its trnode markers.  Tile lists only cull, so frames are unchanged.
 * Needed for snapshots that carry no tile lists (tiling off, synthetic scenes).  The tile size is
 * the snapshot's unless it has a single tile, then 32x8 (engine.h:38-39).
 * Setting the environment variable QR_REBIN=1 turns the flag on for every upload.
 */
#define QR_UPLOAD_REBIN_TILES 1u
int qr_scene_upload_ex(const void *blob, uint64_t size, int device, uint32_t flags, qr_device_scene **out);
int qr_scene_destroy(qr_device_scene *scn);

/*
 * Host-only half of an upload: validate the snapshot and compile it into the device image the kernel walks
 * (contiguous list programs, clipper programs, light lists, wave schedule; csrc/qr_program.h), verify every
 * offset in the image, and report its size.  Needs no GPU: the same code runs inside qr_scene_upload, so a
 * snapshot this call accepts cannot make the kernel read outside the image.
 */
typedef struct qr_program_info
{
    uint64_t bytes;             /* size of the device image                                   */
    uint32_t n_lists;           /* distinct surface lists compiled                            */
    uint32_t n_cells;           /* cells in them (END cells not counted)                      */
    uint32_t n_dropped;         /* snapshot cells that needed no device cell (markers)        */
    uint32_t n_clip_cells;      /* cells of clipper programs                                  */
    uint32_t n_sched;           /* wave-schedule entries (= waves of a whole-frame launch)    */
    uint32_t n_grids;           /* shadow lists by hit position built for large planes        */
    uint32_t n_grid_lists;      /* lists in them                                              */
    uint32_t n_dda;             /* uniform grids built over long lists                        */
} qr_program_info;
int qr_program_stats(const void *blob, uint64_t size, qr_program_info *info);

/*
 * Host-only: build per-surface shadow, reflection / refraction and light lists for a snapshot that carries one
 * global surface list (qr_frame.clist) -- the role of rt_SceneThread::ssort / lsort (engine.cpp:2134-2753) with
 * their bbox_shad / bbox_side culling (rtgeom.cpp:1004, 1954), from the surfaces' conservative bounds.  Lists
 * only cull: frames are unchanged.  Returns a NEW snapshot (release with qr_free) whose surfaces point at the
 * new lists; the global list's order, bounding-volume arrays and trnode markers are preserved in every list.
 */
int qr_snapshot_build_lists_c(const void *blob, uint64_t size, void **out_blob, uint64_t *out_size);
int qr_scene_get_info(const qr_device_scene *scn, qr_scene_info *info);

/* Override recursion depth (s_inf->depth, tracer.h:173) of an uploaded scene. */
int qr_scene_set_depth(qr_device_scene *scn, int depth);

/*
 * Path-tracer mode (the reference's RT_FEAT_PT, tracer.cpp:1112-1136, 1218-1285, 2339-2703, 3428-3466, 5176-5219;
 * rt_Scene::set_pton, engine.cpp:3729).  on != 0 (re)starts the accumulation: per pixel sample one LCG24 state,
 * seeded like rt_Scene::reset_pseed, and three colour planes on the device.  Every qr_render_async then adds ONE sample
 * per pixel sample and writes the running mean as the frame.  on == 2: shade in the reference's (eager) order, which
 * reproduces the reference's frames pixel for pixel (every depth, any number of accumulated frames; slow).  on == 1: the fast kernel with deferred shading: statistically equivalent to the
 * reference, not bit-exact: the reference's stream of random numbers depends on its eager shading order.
 * Rendered by a packet-walk kernel instance of its own; ids / counting renders are refused in this mode.  qr_render0
 * (drop-in) uses the engine's own planes and the eager kernel.
 */
int qr_scene_set_pt(qr_device_scene *scn, int on);

/*
 * Restrict rendering to framebuffer rows [row_begin, row_end) and, inside that,
 * to rows with (y % thnum) == index  -- the reference's thread interleave
 * (tracer.cpp:1144-1145, 5385-5386).  Default: whole frame, index 0, thnum 1.
 * Multi-GPU sharding uses tile-row groups through this call.
 */
int qr_scene_set_rows(qr_device_scene *scn, int row_begin, int row_end, int index, int thnum);

/*
 * Multi-GPU sharding: render only the 8-row tile rows first, first+stride, ...
 * (tile height RT_TILE_H = 8, engine.h:39).  Rank r of N uses (r, N).
 */
int qr_scene_set_tile_rows(qr_device_scene *scn, int first, int stride);

/*
 * Launch the render on `stream` (a hipStream_t passed as void*, NULL = the
 * default stream).  `frame_dev` is DEVICE memory, frm_w*frm_h uint32, compact
 * stride frm_w; rows outside the selected set are left untouched.  Asynchronous.
 */
int qr_render_async(qr_device_scene *scn, void *frame_dev, void *stream);

/*
 * One launch for several row ranges: target i renders rows [row_begin[i], row_end[i]) of scenes[i] into
 * frames_dev[i] (device memory, compact stride as for qr_render_async).  The scenes may be the same or
 * different ones on the same device; at most 16 targets.  This is what a GPU does in a sharded step (its
 * block of every frame in flight, see quadray-engine_amd/sharding.py): issued as separate launches the
 * blocks pay a ramp, a drain and a tail each.  Asynchronous on `stream`.
 */
int qr_render_multi_async(int n, qr_device_scene *const *scenes, void *const *frames_dev,
                          const int *row_begin, const int *row_end, void *stream);

/*
 * As qr_render_async, additionally writing the visible primary hit of every
 * pixel to `ids_dev` (int32 per pixel: surface_index << 1 | side, -1 = none).
 * The reference has no such buffer; it is compared against the oracle's.
 */
int qr_render_ids_async(qr_device_scene *scn, void *frame_dev, void *ids_dev, void *stream);

/*
 * Same, plus per-kind ray counting (slower kernel variant; counts are
 * deterministic for a given scene/depth/rows).  Synchronises the stream.
 */
int qr_render_count(qr_device_scene *scn, void *frame_dev, void *stream, qr_ray_counts *counts);

/*
 * Convenience: render to a HOST frame buffer (stride `row_pixels`, may be
 * negative like the reference's x_row), synchronous.  With several entries in QR_DEVICES a whole-frame call renders one
 * band per entry (the scene image is copied to the other devices on the first such call, peer to peer).
 */
int qr_render_host(qr_device_scene *scn, uint32_t *frame_host, int row_pixels);

/*
 * Time `iters` back-to-back launches with HIP events recorded on `stream`
 * around each launch; returns average/min kernel milliseconds.
 */
int qr_render_timed(qr_device_scene *scn, void *frame_dev, void *stream,
                    int iters, float *avg_ms, float *min_ms);

/* ------------------------------------------------------------------------ */
/* 3. Misc                                                                   */
/* ------------------------------------------------------------------------ */

/*
 * Fingerprint of a frame: FNV-1a-64 over (pixel & 0xFFFFFF) as 4 little-endian bytes, row-major, compact
 * stride.  The same definition tests/golden/manifest.json uses for the reference's frames, so a caller
 * (bench.py) can check a rendered frame against the reference without any test code.
 */
uint64_t qr_frame_hash(const uint32_t *frame_host, uint64_t n_pixels);

const char *qr_last_error(void);
const char *qr_version(void);
int qr_device_count(void);

/* name of the dominant kernel as it appears in rocprofv3 --kernel-trace */
const char *qr_kernel_name(void);

#ifdef __cplusplus
}
#endif

#endif /* QRHIP_H */
