/*
 * qr_sides.h - per-side placement of nodes and lights (the engine's bbox_side / clip_side, rtgeom.cpp:939-995,
 * 1954-2128) restated over a snapshot; see qr_sides.cpp.  Host code, used by the list-building pass.
 */
#ifndef QR_SIDES_H
#define QR_SIDES_H

#include "qr_scene.h"

struct QrSideGeom
{
    struct Box;
    const qr_scene_view &v;
    int n = 0;                  /* surfaces of the snapshot: boxes [0, n) */
    Box *box = nullptr;         /* followed by the boxes of array elements (add_array_box) */
    int n_box = 0, cap_box = 0;

    explicit QrSideGeom(const qr_scene_view &view);
    ~QrSideGeom();
    QrSideGeom(const QrSideGeom &) = delete;
    QrSideGeom &operator=(const QrSideGeom &) = delete;

    /* sides of the clipped surface `srf` the box `ref` (a surface's, or an array's from add_array_box) is seen from:
     * 0 none, 1 inner, 2 outer, 3 both */
    int side(int ref, int srf) const;
    /* box of an array element from the surfaces nested under it; space: trnode record its box lives in, QR_NULL = world */
    int add_array_box(const int *leaves, int n_leaves, int space);
    /* may the box `caster` cast a shadow on the box `receiver` as seen from the point `light` (bbox_shad, rtgeom.cpp:1004) */
    int shad(const float *light, int caster, int receiver) const;
    /* the same for a point (a light): 1 inner, 2 outer, 3 both */
    int clip_side(int srf, const float *pos) const;
    /* does the engine build per-side surface lists for `srf` (some side reflects or is not opaque, engine.cpp:2155-2176) */
    bool builds_side_lists(int srf) const;
    /* bounding sphere of the surface's box as the engine computes it (mid, rad of rt_BOUND); false: unbounded */
    bool box_sphere(int srf, float mid[3], float *rad) const;

private:
    void box_geometry(Box &b) const;
    void node_tran(const Box &b, const float *pos, float *out) const;
    int surf_side(const Box &s, const float *pos) const;
    int surf_cbox(const Box &s, const float *pos) const;
    int node_bbox(const Box &o, const float *pos) const;
    int clip_conc(const Box &s) const;
    int surf_hole(int srf, int ref) const;
    int surf_clip(int srf, int clp) const;
    int bbox_fuse(int i1, int i2) const;
};

#endif /* QR_SIDES_H */
