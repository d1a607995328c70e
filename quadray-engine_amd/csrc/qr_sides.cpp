/*
 * qr_sides.cpp - which side(s) of a clipped surface another object is seen from, restated from the snapshot.
 *
 * The engine places every node of its hierarchy on the OUTER and / or the INNER list of a surface (the lists
 * reflection / refraction rays walk, rt_SIMD_SURFACE::lst_p[1] / lst_p[3]) and every light on the outer and / or inner
 * light list by one predicate, bbox_side (core/engine/rtgeom.cpp:1954-2128, called from rt_SceneThread::ssort / lsort,
 * engine.cpp:2134-2753, under RT_OPTS_2SIDED with the clip-relation rules of RT_OPTS_2SIDED_EXT2 and the box test of
 * RT_OPTS_2SIDED_EXT1).  That placement is part of the engine's PICTURE, not a neutral cull: a convex shell is absent
 * from its own outer list, a bowl's inside only holds what its box test puts there.  This file restates the predicate
 * and everything it rests on over the fields a snapshot carries:
 *
 *   reference (rtgeom.cpp)                         here
 *   rt_BOUND bmin / bmax (sub-world box)           qr_surface.min / max + pos  (rt_Surface::update_bounds stores them
 *                                                  relative to pos, object.cpp:2832-2844)
 *   rt_SHAPE cmin / cmax (clip box)                the same values where qr_surface.minmax_t says the axis is clipped
 *                                                  (min_t / max_t, object.cpp:2824-2830), -inf / +inf elsewhere
 *   verts, mid, rad (rt_Node::update_bbgeom,       box corners through the trnode's forward matrix = the inverse of the
 *     object.cpp:849-1091)                         3x3 the snapshot holds (tci / tcj / tck rows), in double precision
 *   node_tran (775-791)                            the trnode's rows applied to pos - trnode.pos
 *   surf_hole / surf_clip (607-709)                the clipper list of the snapshot (accum markers, trnode elements)
 *   surf_conc / clip_conc (718-766), surf_cbox (802-840), node_bbox (848-885), surf_side (894-931),
 *   clip_side (939-995), vert_face (314-441), bbox_fuse (1853-1943), bbox_side (1954-2128): as they are.
 *
 * What a snapshot cannot say: whether the engine allocated box geometry for a surface (verts_num, decided once from the
 * scene description's axis clippers, e.g. object.cpp:3097-3111); here a surface has box geometry iff its box is finite
 * on every axis (the two only part ways when custom clippers alone make an unclipped shape's box finite,
 * RT_OPTS_ADJUST).  Arrays' own boxes (rt_Array::update_bounds) are not rebuilt: the engine asks bbox_side for an array
 * only to skip the calls for its members when the whole box lies on one side, and a member's box lies inside its
 * array's, so asking every member gives the same placement (tests/test_lists.py compares with the engine's own lists
 * on every fixture).
 */
#include "qr_internal.h"
#include "qr_sides.h"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace {

const float kInf = FLT_MAX;             /* RT_INF, rtbase.h:557 */
const float kThr = 0.0001f;             /* RT_CULL_THRESHOLD, rtgeom.h:33 */

/* bx_edges / bx_faces, object.cpp:680-706 */
const int kEdges[12][2] = { {0, 1}, {1, 2}, {2, 3}, {3, 0}, {0, 4}, {1, 5}, {2, 6}, {3, 7}, {7, 6}, {6, 5}, {5, 4}, {4, 7} };
const int kFaces[6][4] = { {0, 1, 2, 3}, {0, 4, 5, 1}, {1, 5, 6, 2}, {2, 6, 7, 3}, {3, 7, 4, 0}, {7, 6, 5, 4} };

inline bool real_srf(const qr_surface &s) { return s.srf_t[3] >= 0 && s.srf_t[3] < QR_TAG_SURFACE_MAX; }
inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

} // namespace

/* one surface as rtgeom sees it: rt_BOUND + rt_SHAPE */
struct QrSideGeom::Box
{
    bool real = false, plane = false, own = false;      /* own: the surface is its own trnode (pos is in the matrix) */
    bool array = false;                                  /* the box of an array element (add_array_box) */
    int tag = 0, trn = QR_NULL;
    int mp[3] = {0, 1, 2};
    float pps[3] = {0, 0, 0};
    float bmin[3], bmax[3], cmin[3], cmax[3];
    float sci[4], scj[3], sck[3];
    float inv[3][3], tpos[3];                            /* trnode: rows of the inverse 3x3, position */
    int nverts = 0, nedges = 0, nfaces = 0;
    float verts[8][4];
    int edge_k[12], face_k[6], face_i[6], face_j[6];
    float mid[3] = {0, 0, 0}, rad = kInf;
    bool can_see = false;                                /* some side's material reflects or is not opaque */
};

QrSideGeom::~QrSideGeom() { delete[] box; }

/* corners, edges and faces of a finite box, centre and radius of its sphere: rt_Node::update_bbgeom, object.cpp:849-1091 */
void QrSideGeom::box_geometry(Box &b) const
{
    const int mi = b.mp[0], mj = b.mp[1], mk = b.mp[2];
    const int nv = b.plane ? 4 : 8;
    static const int hi_i[8] = {1, 0, 0, 1, 1, 0, 0, 1}, hi_j[8] = {1, 1, 0, 0, 1, 1, 0, 0}, hi_k[8] = {1, 1, 1, 1, 0, 0, 0, 0};
    double F[3][3] = { {1, 0, 0}, {0, 1, 0}, {0, 0, 1} };
    if (b.trn != QR_NULL)
    {
        /* forward 3x3 of the trnode = inverse of the rows the snapshot holds */
        const float (*m)[3] = b.inv;
        const double a00 = m[0][0], a01 = m[0][1], a02 = m[0][2], a10 = m[1][0], a11 = m[1][1], a12 = m[1][2], a20 = m[2][0], a21 = m[2][1], a22 = m[2][2];
        const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
        const double r = det != 0.0 ? 1.0 / det : 0.0;
        F[0][0] = (a11 * a22 - a12 * a21) * r; F[0][1] = (a02 * a21 - a01 * a22) * r; F[0][2] = (a01 * a12 - a02 * a11) * r;
        F[1][0] = (a12 * a20 - a10 * a22) * r; F[1][1] = (a00 * a22 - a02 * a20) * r; F[1][2] = (a02 * a10 - a00 * a12) * r;
        F[2][0] = (a10 * a21 - a11 * a20) * r; F[2][1] = (a01 * a20 - a00 * a21) * r; F[2][2] = (a00 * a11 - a01 * a10) * r;
    }
    for (int k = 0; k < nv; k++)
    {
        float c[3];
        c[mi] = hi_i[k] ? b.bmax[mi] : b.bmin[mi];
        c[mj] = hi_j[k] ? b.bmax[mj] : b.bmin[mj];
        c[mk] = hi_k[k] ? b.bmax[mk] : b.bmin[mk];
        if (b.trn != QR_NULL)
            for (int a = 0; a < 3; a++)
                b.verts[k][a] = (float)(F[a][0] * c[0] + F[a][1] * c[1] + F[a][2] * c[2] + (double)b.tpos[a]);
        else
            for (int a = 0; a < 3; a++) b.verts[k][a] = c[a];
        b.verts[k][3] = 1.0f;
    }
    b.nverts = nv; b.nedges = b.plane ? 4 : 12; b.nfaces = b.plane ? 1 : 6;
    const bool aligned = b.trn == QR_NULL;
    static const int ek[12] = {0, 1, 0, 1, 2, 2, 2, 2, 0, 1, 0, 1};                /* edge directions as local axes i / j / k */
    static const int fk[6] = {2, 1, 0, 1, 0, 2}, fi[6] = {0, 2, 2, 2, 2, 0}, fj[6] = {1, 0, 1, 0, 1, 1};
    for (int e = 0; e < 12; e++) b.edge_k[e] = aligned ? b.mp[ek[e]] : 3;
    for (int f = 0; f < 6; f++)
    {
        b.face_k[f] = aligned ? b.mp[fk[f]] : 3; b.face_i[f] = aligned ? b.mp[fi[f]] : 3; b.face_j[f] = aligned ? b.mp[fj[f]] : 3;
    }
    const float f = 1.0f / (float)nv;
    b.mid[0] = b.mid[1] = b.mid[2] = 0.0f;
    for (int k = 0; k < nv; k++) for (int a = 0; a < 3; a++) b.mid[a] += b.verts[k][a] * f;
    float rad = 0.0f;
    for (int k = 0; k < nv; k++)
    {
        const float d[3] = { b.mid[0] - b.verts[k][0], b.mid[1] - b.verts[k][1], b.mid[2] - b.verts[k][2] };
        const float dd = dot3(d, d);
        if (rad < dd) rad = dd;
    }
    b.rad = sqrtf(rad);
}

/*
 * The box of an ARRAY element of the hierarchy (rt_Array::update_bounds, object.cpp:1830-2320: bvbox of a bounding-volume
 * node in world space, trbox / inbox of a transform node in its own space) from the surfaces nested under it: their boxes
 * taken as they are when they live in the box's space (`space`: index of the trnode record, QR_NULL = world), their
 * world-space corners otherwise.  The engine unites sub-arrays level by level; uniting the leaves gives the same box except
 * under nested transform nodes (a transformed sub-array contributes the corners of ITS box, a looser fit).  An unbounded
 * member leaves the array without box geometry, as in the engine (rad stays RT_INF).
 */
int QrSideGeom::add_array_box(const int *leaves, int n_leaves, int space)
{
    if (n_box == cap_box)
    {
        const int cap = cap_box * 2 + 16;
        Box *nb = new Box[(size_t)cap];
        for (int i = 0; i < n_box; i++) nb[i] = box[i];
        delete[] box; box = nb; cap_box = cap;
    }
    Box &b = box[n_box];
    b = Box();
    b.real = true; b.array = true; b.tag = QR_TAG_ARRAY; b.trn = space;
    if (space != QR_NULL)
    {
        const qr_surface &t = v.srf[space];
        for (int a = 0; a < 3; a++) { b.inv[0][a] = t.tci[a]; b.inv[1][a] = t.tcj[a]; b.inv[2][a] = t.tck[a]; b.tpos[a] = t.pos[a]; }
    }
    for (int a = 0; a < 3; a++) { b.bmin[a] = +kInf; b.bmax[a] = -kInf; b.cmin[a] = -kInf; b.cmax[a] = +kInf; }
    bool bounded = n_leaves > 0;
    for (int k = 0; k < n_leaves && bounded; k++)
    {
        const Box &m = box[leaves[k]];
        if (m.rad == kInf || m.nverts == 0) { bounded = false; break; }
        if (m.trn == space)
            for (int a = 0; a < 3; a++) { b.bmin[a] = fminf(b.bmin[a], m.bmin[a]); b.bmax[a] = fmaxf(b.bmax[a], m.bmax[a]); }
        else if (space == QR_NULL)
            for (int q = 0; q < m.nverts; q++)
                for (int a = 0; a < 3; a++) { b.bmin[a] = fminf(b.bmin[a], m.verts[q][a]); b.bmax[a] = fmaxf(b.bmax[a], m.verts[q][a]); }
        else bounded = false;           /* a member of another transform node inside this one's box: not a case the engine's lists have */
    }
    if (bounded) box_geometry(b);
    return n_box++;
}

QrSideGeom::QrSideGeom(const qr_scene_view &view) : v(view)
{
    n = (int)v.hdr->n_srf;
    box = new Box[(size_t)n];
    n_box = cap_box = n;
    for (int i = 0; i < n; i++)
    {
        const qr_surface &s = v.srf[i];
        Box &b = box[i];
        b.real = real_srf(s);
        if (!b.real) continue;
        b.tag = s.srf_t[3];
        b.plane = b.tag == QR_TAG_PLANE;
        b.trn = s.trnode;
        b.own = s.trnode == i;
        for (int a = 0; a < 3; a++) b.mp[a] = (int)((s.axes >> (2 * a)) & 3);
        for (int a = 0; a < 3; a++) b.pps[a] = b.own ? 0.0f : s.pos[a];
        for (int a = 0; a < 3; a++)
        {
            /* update_bounds stored bmin - pps; an unbounded axis holds -/+RT_INF (the subtraction does not move it) */
            b.bmin[a] = s.min[a] <= -kInf ? -kInf : s.min[a] + b.pps[a];
            b.bmax[a] = s.max[a] >= +kInf ? +kInf : s.max[a] + b.pps[a];
            b.cmin[a] = (s.minmax_t & (1u << a)) ? b.bmin[a] : -kInf;
            b.cmax[a] = (s.minmax_t & (1u << (3 + a))) ? b.bmax[a] : +kInf;
        }
        for (int a = 0; a < 4; a++) b.sci[a] = s.sci[a];
        /* the snapshot keeps half of the shape's linear coefficients (rt_Quadric::commit_fields, object.cpp:3059-3061) */
        for (int a = 0; a < 3; a++) b.scj[a] = s.scj[a] + s.scj[a];
        for (int a = 0; a < 3; a++) b.sck[a] = 0.0f;
        b.sck[b.mp[2]] = ((s.axes >> 10) & 1) ? -1.0f : 1.0f;                 /* the plane's normal: axis k with its sign */
        if (b.trn != QR_NULL)
        {
            const qr_surface &t = v.srf[b.trn];
            for (int a = 0; a < 3; a++) { b.inv[0][a] = t.tci[a]; b.inv[1][a] = t.tcj[a]; b.inv[2][a] = t.tck[a]; b.tpos[a] = t.pos[a]; }
        }
        b.can_see = ((s.props[0] | s.props[1]) & QR_PROP_REFLECT) != 0 || (s.props[0] & QR_PROP_OPAQUE) == 0 || (s.props[1] & QR_PROP_OPAQUE) == 0;

        /* box geometry, rt_Node::update_bbgeom: 4 corners of a plane's rectangle, 8 of a box */
        bool finite = true;
        for (int a = 0; a < 3; a++) finite = finite && b.bmin[a] > -kInf && b.bmax[a] < kInf;
        if (!finite) continue;
        box_geometry(b);
    }
}

/* node_tran, rtgeom.cpp:775-791: pos in the trnode's sub-world space */
void QrSideGeom::node_tran(const Box &b, const float *pos, float *out) const
{
    if (b.trn == QR_NULL) { out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2]; return; }
    const float d[3] = { pos[0] - b.tpos[0], pos[1] - b.tpos[1], pos[2] - b.tpos[2] };
    /* matrix_mul_vector, rtgeom.cpp:59-77: four products summed left to right, the last one with dff[W] = 0 */
    for (int a = 0; a < 3; a++) out[a] = b.inv[a][0] * d[0] + b.inv[a][1] * d[1] + b.inv[a][2] * d[2] + 0.0f;
}

/* surf_side 894-931: 0 on the surface (with margin), 1 inner, 2 outer */
int QrSideGeom::surf_side(const Box &s, const float *pos) const
{
    float p[3], loc[3];
    node_tran(s, pos, p);
    for (int a = 0; a < 3; a++) loc[a] = s.own ? p[a] : p[a] - s.pps[a];
    float d;
    if (s.plane) d = dot3(loc, s.sck);
    else
    {
        const float dcj = dot3(loc, s.scj);
        const float dci = loc[0] * loc[0] * s.sci[0] + loc[1] * loc[1] * s.sci[1] + loc[2] * loc[2] * s.sci[2];
        d = dci - dcj - s.sci[3];
    }
    return d > (0.0f + kThr) ? 2 : d >= (0.0f - kThr) ? 0 : 1;
}

/* surf_cbox 802-840: 0 inside the clip box, 1 outside, 2 on its border (all with margin) */
int QrSideGeom::surf_cbox(const Box &s, const float *pos) const
{
    float p[3];
    node_tran(s, pos, p);
    bool out = false, edge = false;
    for (int a = 0; a < 3; a++) out = out || p[a] + kThr < s.cmin[a];
    for (int a = 0; a < 3; a++) out = out || p[a] - kThr > s.cmax[a];
    if (out) return 1;
    for (int a = 0; a < 3; a++) edge = edge || p[a] - kThr <= s.cmin[a];
    for (int a = 0; a < 3; a++) edge = edge || p[a] + kThr >= s.cmax[a];
    return edge ? 2 : 0;
}

/* node_bbox 848-885: 1 inside the box, 2 on its border, 0 outside */
int QrSideGeom::node_bbox(const Box &o, const float *pos) const
{
    float p[3];
    node_tran(o, pos, p);
    bool in = true, on = true;
    for (int a = 0; a < 3; a++) in = in && p[a] - kThr > o.bmin[a] && p[a] + kThr < o.bmax[a];
    if (in) return 1;
    for (int a = 0; a < 3; a++) on = on && p[a] + kThr >= o.bmin[a] && p[a] - kThr <= o.bmax[a];
    return on ? 2 : 0;
}

/* surf_conc 718-732 / clip_conc 741-766 */
static inline bool concave_tag(int tag) { return tag == 3 || tag == 5 || tag == 7 || tag == 8; }    /* cone, hyperboloid, hypercylinder, hyperparaboloid */
int QrSideGeom::clip_conc(const Box &s) const
{
    const int k = s.mp[2];
    if ((s.tag == 3 || s.tag == 5 || s.tag == 7)
        && ((s.sci[3] <= 0.0f && s.bmin[k] < s.pps[k] && s.bmax[k] > s.pps[k]) || s.sci[3] > 0.0f)) return 1;
    return s.tag == 8 ? 1 : 0;
}

/* surf_hole 607-660: 1 axis clippers cut the shape, 2 a custom clipper other than `ref` (or one inside an accum segment) */
int QrSideGeom::surf_hole(int srf, int ref) const
{
    const Box &s = box[srf];
    if (s.plane) return 0;
    int c = 0;
    for (int a = 0; a < 3; a++) if (s.cmin[a] != -kInf || s.cmax[a] != +kInf) c |= 1;
    int skip = 0;
    for (int e = v.srf[srf].clip; e != QR_NULL; e = v.elm[e].next)
    {
        const qr_elem &el = v.elm[e];
        if (el.simd == QR_NULL) { skip = 1 - skip; continue; }         /* accum marker */
        if (el.kind == 2 || !real_srf(v.srf[el.simd])) continue;       /* trnode element */
        if (el.simd != ref || skip == 1) { c |= 2; break; }
    }
    return c;
}

/* surf_clip 669-709: does `clp` clip `srf` outside of any accum segment: 0 no, 1 by its inner side, 2 by its outer side */
int QrSideGeom::surf_clip(int srf, int clp) const
{
    int c = 0, skip = 0;
    for (int e = v.srf[srf].clip; e != QR_NULL; e = v.elm[e].next)
    {
        const qr_elem &el = v.elm[e];
        if (el.simd == QR_NULL) { skip = 1 - skip; continue; }
        if (el.kind == 2 || !real_srf(v.srf[el.simd])) continue;
        if (el.simd == clp && skip == 0) { c = el.data; break; }
    }
    return c == 0 ? 0 : 1 + ((1 + c) >> 1);
}

/* clip_side 939-995: which side of the clipped surface a POINT (a light, the camera) sees: 1 inner, 2 outer, 3 both */
int QrSideGeom::clip_side(int srf, const float *pos) const
{
    const Box &s = box[srf];
    int c = surf_side(s, pos);
    if (c == 0) return 3;
    if (s.plane) return c;
    if (!concave_tag(s.tag) && c == 1) return c;
    const int k = surf_hole(srf, srf);
    if (k == 0) return c;
    if (k & 2) return 3;
    return surf_cbox(s, pos) != 0 ? 3 : c;
}

/*
 * Geometry helpers of the two crossing tests below.  The predicates must agree with the engine's fp32 results bit for bit
 * (tests/test_lists.py: every list of 43 scenes member for member), which fixes the ORDER of the arithmetic -- differences,
 * cross products with rtgeom.h:135-141's operand order, left-to-right sums, a tolerance of kThr on parameters and of th * kThr
 * on windows -- but not how the code is cut: here a crossing test is "a parameter num / den with den >= 0" plus "windows the
 * other coordinates must fall into", and both tests share the classification of that parameter.
 */
struct F3 { float x, y, z; };
static inline F3 f3(const float *p) { F3 r = { p[0], p[1], p[2] }; return r; }
static inline F3 f3_sub(const float *a, const float *b) { F3 r = { a[0] - b[0], a[1] - b[1], a[2] - b[2] }; return r; }
static inline F3 f3_cross(const F3 &a, const F3 &b) { F3 r = { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; return r; }
static inline float f3_dot(const F3 &a, const F3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float flip_if(float v, bool neg) { return neg ? -v : v; }

/* is v outside [lo, hi] once both ends are scaled by den (den >= 0)? */
static inline bool off_window(float v, float lo, float hi, float den) { return v < lo * den || v > hi * den; }

/* where num / den (den >= 0) lies on the unit interval: 1 beyond its end, 3 at the end, 2 strictly inside, 4 at the start,
 * 0 before it -- the codes of the engine's vert_face / edge_edge (rtgeom.cpp:436-440, 587-591) */
static inline int unit_zone(float num, float den)
{
    if (num > (1.0f + kThr) * den) return 1;
    if (num >= (1.0f - kThr) * den) return 3;
    if (num > (0.0f + kThr) * den) return 2;
    if (num >= (0.0f - kThr) * den) return 4;
    return 0;
}

/* the segment a-b against an axis-aligned rectangle: normal axis k, spanned along i by c0-c1 and along j by c0-c2 */
static int seg_rect_aligned(const float *a, const float *b, int th, const float *c0, const float *c1, const float *c2, int k, int i, int j)
{
    const float slack = (float)th * kThr;
    float den = b[k] - a[k];
    const float num = flip_if(c0[k] - a[k], den < 0.0f);
    den = fabsf(den);
    const float at_i = (b[i] - a[i]) * num;
    if (off_window(at_i, fminf(c0[i], c1[i]) - a[i] - slack, fmaxf(c0[i], c1[i]) - a[i] + slack, den)) return 0;
    const float at_j = (b[j] - a[j]) * num;
    if (off_window(at_j, fminf(c0[j], c2[j]) - a[j] - slack, fmaxf(c0[j], c2[j]) - a[j] + slack, den)) return 0;
    return unit_zone(num, den);
}

/* ... and against a parallelogram in general position (corner c0, edges to c1 and c2): barycentric windows [0, 1] */
static int seg_quad(const float *a, const float *b, int th, const float *c0, const float *c1, const float *c2)
{
    const float slack = (float)th * kThr;
    const F3 e1 = f3_sub(c1, c0), e2 = f3_sub(c2, c0), dir = f3_sub(b, a), rel = f3_sub(a, c0);
    const F3 m = f3_cross(dir, e2);
    float den = f3_dot(e1, m);
    const bool neg = den < 0.0f;
    den = fabsf(den);
    if (off_window(flip_if(f3_dot(rel, m), neg), 0.0f - slack, 1.0f + slack, den)) return 0;
    const F3 n = f3_cross(rel, e1);
    if (off_window(flip_if(f3_dot(dir, n), neg), 0.0f - slack, 1.0f + slack, den)) return 0;
    return unit_zone(flip_if(f3_dot(e2, n), neg), den);
}

/* vert_face (rtgeom.cpp:314-441): where along the segment p0-p1 it passes through the face q0-q1-q2; qk / qi / qj < 3 name the
 * axes of an axis-aligned face */
static int vert_face(const float *p0, const float *p1, int th, const float *q0, const float *q1, const float *q2, int qk, int qi, int qj)
{
    return (qk < 3 && qi < 3 && qj < 3) ? seg_rect_aligned(p0, p1, th, q0, q1, q2, qk, qi, qj) : seg_quad(p0, p1, th, q0, q1, q2);
}

/* bbox_fuse 1853-1943: 0 apart, 1 possibly one inside the other, 2 borders intersect (or no bounds to tell) */
int QrSideGeom::bbox_fuse(int i1, int i2) const
{
    const Box &a = box[i1], &b = box[i2];
    if (a.rad == kInf || b.rad == kInf || i1 == i2) return 2;
    const float d[3] = { a.mid[0] - b.mid[0], a.mid[1] - b.mid[1], a.mid[2] - b.mid[2] };
    if (a.rad + b.rad < sqrtf(dot3(d, d))) return 0;
    if (a.nverts == 0 || b.nverts == 0) return 1;
    if (node_bbox(a, b.mid) != 0) return 1;
    if (node_bbox(b, a.mid) != 0) return 1;
    for (int pass = 0; pass < 2; pass++)
    {
        const Box &x = pass == 0 ? a : b, &y = pass == 0 ? b : a;
        for (int e = 0; e < x.nedges; e++)
            for (int f = 0; f < y.nfaces; f++)
                if (vert_face(x.verts[kEdges[e][0]], x.verts[kEdges[e][1]], +1,
                              y.verts[kFaces[f][0]], y.verts[kFaces[f][1]], y.verts[kFaces[f][3]],
                              y.face_k[f], y.face_i[f], y.face_j[f]) == 2) return 2;
    }
    return 0;
}

/* bbox_side 1954-2128 for a SURFACE `ref`'s box seen against the clipped surface `srf`: 0 none, 1 inner, 2 outer, 3 both */
int QrSideGeom::side(int ref, int srf) const
{
    const Box &o = box[ref], &s = box[srf];
    int c = 0;
    const int p = s.plane ? 1 : 0;
    const int k = surf_hole(srf, o.array ? QR_NULL : ref);
    const int m = concave_tag(s.tag) ? 1 : 0;

    /* clip relations between two surfaces, RT_OPTS_2SIDED_EXT2 (not for the box of an array) */
    if (!o.array)
    {
        if (ref == srf)
        {
            if (p == 0) { c |= 1; if (clip_conc(o) == 1) c |= 2; }
            return c;
        }
        const int i = surf_clip(ref, srf), j = surf_clip(srf, ref);
        if ((i == 2 && j == 2) || (i == 2 && j == 0)) { c |= 1; if (m == 1 && k != 0) c |= 2; return c; }
        if (i == 2 && j == 1) { c |= 1; if (m == 1) c |= 2; return c; }
        if (i == 1 && j == 2) { c |= 2; if (p == 0 && (concave_tag(o.tag) || k != 0)) c |= 1; return c; }
        if (i == 1 && j == 1) { c |= 2; if (p == 0) c |= 1; return c; }
        if (i == 1 && j == 0) { c |= 2; if (p == 0 && k != 0) c |= 1; return c; }
        if ((i == 0 && j == 2) || (i == 0 && j == 1)) return 3;
    }

    /* a plane: the sides its corners lie on */
    if (p == 1)
    {
        if (o.nverts == 0) return 3;
        for (int q = 0; q < o.nverts && c != 3; q++) c |= surf_side(s, o.verts[q]);
        return c;
    }
    int nf = bbox_fuse(ref, srf);
    if ((nf != 0 && m == 1) || nf == 2) return 3;
    if (nf == 1 && m == 0)
    {
        c |= 1;
        for (int q = 0; q < o.nverts; q++) if (surf_side(s, o.verts[q]) == 2) { c |= 2; break; }
        return c;
    }
    if (k == 0) return 2;
    if (k & 2) return 3;
    c |= 2;
    for (int q = 0; q < o.nverts; q++) if (surf_cbox(s, o.verts[q]) != 0) { c |= 1; break; }
    return c;
}

/* two axis-aligned edges seen from `eye`: the first runs along axis pa from a1 to a2, the second along qa from b1 to b2; the
 * third axis carries the depth order */
static int edges_aligned(const float *eye, int th, const float *a1, const float *a2, int pa, const float *b1, const float *b2, int qa)
{
    if (pa == qa) return 0;                                 /* parallel edges never cross */
    const float slack = (float)th * kThr;
    const int depth_axis = 3 - pa - qa;
    float near_d = a1[depth_axis] - eye[depth_axis];
    float far_d = b1[depth_axis] - eye[depth_axis];
    near_d = flip_if(near_d, far_d < 0.0f);
    far_d = fabsf(far_d);
    const float at_p = (b1[pa] - eye[pa]) * near_d;
    if (off_window(at_p, fminf(a1[pa], a2[pa]) - eye[pa] - slack, fmaxf(a1[pa], a2[pa]) - eye[pa] + slack, far_d)) return 0;
    far_d = flip_if(far_d, near_d < 0.0f);
    near_d = fabsf(near_d);
    const float at_q = (a1[qa] - eye[qa]) * far_d;
    if (off_window(at_q, fminf(b1[qa], b2[qa]) - eye[qa] - slack, fmaxf(b1[qa], b2[qa]) - eye[qa] + slack, near_d)) return 0;
    return unit_zone(far_d, near_d);
}

/* ... and two edges in general position */
static int edges_general(const float *eye, int th, const float *a1, const float *a2, const float *b1, const float *b2)
{
    const float slack = (float)th * kThr;
    const F3 ea = f3_sub(a2, a1), eb = f3_sub(b2, b1), ra = f3_sub(a1, eye), rb = f3_sub(b1, eye);
    const F3 m = f3_cross(eb, ea), n = f3_cross(rb, ra);
    float far_d = f3_dot(rb, m);
    bool neg = far_d < 0.0f;
    far_d = fabsf(far_d);
    if (off_window(flip_if(f3_dot(eb, n), neg), 0.0f - slack, 1.0f + slack, far_d)) return 0;
    far_d = flip_if(far_d, neg);
    float near_d = f3_dot(ra, m);
    neg = near_d < 0.0f;
    near_d = fabsf(near_d);
    if (off_window(flip_if(f3_dot(ea, n), neg), 0.0f - slack, 1.0f + slack, near_d)) return 0;
    far_d = flip_if(far_d, neg);
    return unit_zone(far_d, near_d);
}

/* edge_edge (rtgeom.cpp:449-592): do the edges p1-p2 and q1-q2 cross as seen from p0, and in which order (1: the first edge lies
 * between p0 and the second); pk / qk < 3 name the axis of an axis-aligned edge */
static int edge_edge(const float *p0, int th, const float *p1, const float *p2, int pk, const float *q1, const float *q2, int qk)
{
    return (pk < 3 && qk < 3) ? edges_aligned(p0, th, p1, p2, pk, q1, q2, qk) : edges_general(p0, th, p1, p2, q1, q2);
}

static inline float asin32(float a) { return a <= -1.0f ? -(float)(3.14159265358979323846 / 2.0) : a >= 1.0f ? (float)(3.14159265358979323846 / 2.0) : asinf(a); }
static inline float acos32(float a) { return a <= -1.0f ? (float)3.14159265358979323846 : a >= 1.0f ? 0.0f : acosf(a); }
static inline float len3(const float *a) { const float d = dot3(a, a); return d <= 0.0f ? 0.0f : sqrtf(d); }

/* half-angle of the cone from `eye` around a sphere of radius r at distance l (the whole space when the eye is inside) */
static inline float cone_half_angle(float l, float r)
{
    return (l >= r && l > kThr) ? asin32(r / l) : (float)(2.0 * 3.14159265358979323846);
}

/* bbox_shad (rtgeom.cpp:1004-1153): may the box `i1` cast a shadow on the box `i2` as seen from the point `pps` (a light).
 * Three stages, each able to say "no": the cones around the two bounding spheres do not overlap; the caster's sphere lies
 * clear behind the receiver's; no corner, face or edge of one box lines up with the other. */
int QrSideGeom::shad(const float *pps, int i1, int i2) const
{
    const Box &caster = box[i1], &recv = box[i2];
    if (caster.rad == kInf || recv.rad == kInf || i1 == i2) return 1;
    /* clip relations between two surfaces, RT_OPTS_SHADOW_EXT2 */
    if (!caster.array && !recv.array && (surf_clip(i2, i1) != 0 || surf_clip(i1, i2) != 0)) return 1;

    /* stage 1: view cones */
    const float to_c[3] = { caster.mid[0] - pps[0], caster.mid[1] - pps[1], caster.mid[2] - pps[2] };
    const float to_r[3] = { recv.mid[0] - pps[0], recv.mid[1] - pps[1], recv.mid[2] - pps[2] };
    const float dist_c = len3(to_c), dist_r = len3(to_r);
    float cosine = dot3(to_c, to_r);
    cosine = dist_c <= kThr ? 0.0f : cosine / dist_c;
    const float half_c = cone_half_angle(dist_c, caster.rad);
    cosine = dist_r <= kThr ? 0.0f : cosine / dist_r;
    const float half_r = cone_half_angle(dist_r, recv.rad);
    if (half_c + half_r < acos32(cosine)) return 0;

    /* stage 2: spheres apart and the caster the farther one */
    const float apart[3] = { caster.mid[0] - recv.mid[0], caster.mid[1] - recv.mid[1], caster.mid[2] - recv.mid[2] };
    if (caster.rad + recv.rad < len3(apart) && dist_c > dist_r) return 0;

    /* stage 3: box geometry, RT_OPTS_SHADOW_EXT1 */
    if (caster.nverts == 0 || recv.nverts == 0) return 1;
    if (node_bbox(caster, pps) != 0) return 1;
    /* a sight line to a corner of one box through a face of the other: the caster's corner in front of the receiver's face
     * (zone 1: the face lies beyond the corner), or the receiver's corner behind the caster's face (zones 2 / 4) */
    auto face_zone = [&](const float *corner, const Box &b, int f) {
        return vert_face(pps, corner, +1, b.verts[kFaces[f][0]], b.verts[kFaces[f][1]], b.verts[kFaces[f][3]], b.face_k[f], b.face_i[f], b.face_j[f]);
    };
    for (int q = 0; q < caster.nverts; q++)
        for (int f = 0; f < recv.nfaces; f++)
            if (face_zone(caster.verts[q], recv, f) == 1) return 1;
    for (int q = 0; q < recv.nverts; q++)
        for (int f = 0; f < caster.nfaces; f++)
        {
            const int z = face_zone(recv.verts[q], caster, f);
            if (z == 2 || z == 4) return 1;
        }
    /* silhouettes crossing with the caster's edge in front */
    for (int e = 0; e < caster.nedges; e++)
        for (int g = 0; g < recv.nedges; g++)
            if (edge_edge(pps, +1, caster.verts[kEdges[e][0]], caster.verts[kEdges[e][1]], caster.edge_k[e],
                          recv.verts[kEdges[g][0]], recv.verts[kEdges[g][1]], recv.edge_k[g]) == 1) return 1;
    return 0;
}

bool QrSideGeom::builds_side_lists(int srf) const { return box[srf].real && box[srf].can_see; }

bool QrSideGeom::box_sphere(int srf, float mid[3], float *rad) const
{
    const Box &b = box[srf];
    if (!b.real || b.rad == kInf) return false;
    mid[0] = b.mid[0]; mid[1] = b.mid[1]; mid[2] = b.mid[2]; *rad = b.rad;
    return true;
}
