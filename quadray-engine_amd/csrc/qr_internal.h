/*
 * qr_internal.h - declarations shared by the host-side translation units of
 * libqrhip (walker, snapshot I/O, scene compiler, C-ABI glue).  Not part of the public ABI.
 */
#ifndef QR_INTERNAL_H
#define QR_INTERNAL_H

#include "qrhip.h"
#include "qr_scene.h"

#include <string>
#include <vector>
#include <cstdint>

/* flatten the rt_SIMD_INFOX graph into a qr_scene.h blob (qr_walker.cpp) */
struct QrFlattenMap { std::vector<uint64_t> srf, lgt; };    /* address of the engine's record, by snapshot index */
int qr_flatten_impl(const void *s_inf, const qr_abi_desc *abi,
                    std::vector<uint8_t> &out, std::string &err, QrFlattenMap *map = nullptr);

/* thread-local error channel behind qr_last_error() (qr_capi_host.cpp) */
void qr_set_error(const std::string &msg);
int  qr_fail(int status, const std::string &msg);

/* ---- scene compiler (qr_compile.cpp): snapshot -> device image of qr_program.h ---- */

/* conservative world-space bounds of a surface's visible part: sphere (r = +inf: unbounded) and, where the sphere is bounded,
 * the axis-aligned box of the same part (lo > hi: none) */
struct BSphere { float c[3]; float r; float lo[3] = { 1.0f, 1.0f, 1.0f }, hi[3] = { 0.0f, 0.0f, 0.0f }; };

#define QR_SCHED_PER_LANE 0xFFFFFFFEu       /* schedule entry: the footprint straddles tiles, look the list up per pixel */

struct QrProgramStats { uint64_t bytes; uint32_t n_lists, n_cells, n_dropped, n_clip_cells, n_grids, n_grid_lists, n_dda; };

struct QrProgram
{
    std::vector<uint8_t> blob;      /* the device image, offsets relative to its first byte */
    std::vector<uint32_t> order;    /* host copy of the whole-frame wave schedule, 2 words per wave */
    qr_frame frm;
    uint32_t off_order = 0, n_sched = 0;
    uint32_t off_srf = 0, off_shade = 0, off_mat = 0, off_lgt = 0, off_tex = 0, off_tiles = 0, off_lists = 0;
    uint32_t n_srf = 0, n_mat = 0, n_lgt = 0, n_tex = 0, n_tiles = 0;
    QrProgramStats stats = {};
    bool has_grids = false;         /* some light-list entry points at a CGrid: needs the same kernel instance */
    bool has_long_lists = false;    /* some list is flagged QR_LISTF_LONG: the launch uses the kernel instance with the per-lane walk */
    /* sched_blocks > 1: the schedule is grouped by horizontal block of the frame (heavy footprints first inside
     * every block); entries [block_first[k], block_first[k+1]) render rows [block_row[k], block_row[k+1]) */
    std::vector<uint32_t> block_first, block_row;
};

int  qr_snapshot_validate(const qr_scene_view &v, std::string &err);
void qr_bound_spheres(const qr_scene_view &v, std::vector<BSphere> &out);
/* E / T: list cells and tile heads (the snapshot's, or the ones the binning pass built); frm: frame record to use */
int  qr_program_build(const qr_scene_view &v, const std::vector<qr_elem> &E, const std::vector<int32_t> &T,
                      const qr_frame &frm, const std::vector<BSphere> &bs, QrProgram &out, std::string &err, int sched_blocks = 0,
                      bool verify = true);      /* verify: walk every offset of the finished image (qr_program_verify) before returning it */
int  qr_program_verify(const QrProgram &p, std::string &err);
/* per-surface shadow / reflection / light lists from the global list and the surfaces' bounds (ssort / lsort's role) */
int  qr_snapshot_build_lists(const qr_scene_view &v, std::vector<uint8_t> &out, std::string &err);

#endif /* QR_INTERNAL_H */
