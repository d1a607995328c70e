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

/* vert_face 314-441: does the segment p0-p1 meet the quad q0-q1-q2 (two edges from q0); 2 = strictly between */
static int vert_face(const float *p0, const float *p1, int th, const float *q0, const float *q1, const float *q2, int qk, int qi, int qj)
{
    float d, s, t, u, w;
    if (qk < 3 && qi < 3 && qj < 3)
    {
        d = p1[qk] - p0[qk];
        t = q0[qk] - p0[qk];
        t = d < 0.0f ? -t : +t;
        d = fabsf(d);
        u = (p1[qi] - p0[qi]) * t;
        if (u < (fminf(q0[qi], q1[qi]) - p0[qi] - (float)th * kThr) * d || u > (fmaxf(q0[qi], q1[qi]) - p0[qi] + (float)th * kThr) * d) return 0;
        w = (p1[qj] - p0[qj]) * t;
        if (w < (fminf(q0[qj], q2[qj]) - p0[qj] - (float)th * kThr) * d || w > (fmaxf(q0[qj], q2[qj]) - p0[qj] + (float)th * kThr) * d) return 0;
    }
    else
    {
        float e1[3], e2[3], pr[3], qr[3], mx[3], nx[3];
        for (int a = 0; a < 3; a++) { e1[a] = q1[a] - q0[a]; e2[a] = q2[a] - q0[a]; pr[a] = p1[a] - p0[a]; qr[a] = p0[a] - q0[a]; }
        /* RT_VEC3_MUL, rtgeom.h:135-141 */
        mx[0] = pr[1] * e2[2] - e2[1] * pr[2]; mx[1] = pr[2] * e2[0] - e2[2] * pr[0]; mx[2] = pr[0] * e2[1] - e2[0] * pr[1];
        d = dot3(e1, mx);
        s = d < 0.0f ? -1.0f : +1.0f;
        d = fabsf(d);
        u = dot3(qr, mx) * s;
        if (u < (0.0f - (float)th * kThr) * d || u > (1.0f + (float)th * kThr) * d) return 0;
        nx[0] = qr[1] * e1[2] - e1[1] * qr[2]; nx[1] = qr[2] * e1[0] - e1[2] * qr[0]; nx[2] = qr[0] * e1[1] - e1[0] * qr[1];
        w = dot3(pr, nx) * s;
        if (w < (0.0f - (float)th * kThr) * d || w > (1.0f + (float)th * kThr) * d) return 0;
        t = dot3(e2, nx) * s;
    }
    return t > (1.0f + kThr) * d ? 1 : t >= (1.0f - kThr) * d ? 3 : t > (0.0f + kThr) * d ? 2 : t >= (0.0f - kThr) * d ? 4 : 0;
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

/* edge_edge 449-592: do the edges p1-p2 and q1-q2 cross as seen from p0; 1 = the first edge lies between p0 and the second */
static int edge_edge(const float *p0, int th, const float *p1, const float *p2, int pk, const float *q1, const float *q2, int qk)
{
    float d, s, t, u, w;
    if (pk < 3 && qk < 3)
    {
        if (pk == qk) return 0;
        static const int mp[3][3] = { {0, 2, 1}, {2, 1, 0}, {1, 0, 2} };
        const int kk = mp[pk][qk];
        d = p1[kk] - p0[kk];
        t = q1[kk] - p0[kk];
        d = t < 0.0f ? -d : +d;
        t = fabsf(t);
        u = (q1[pk] - p0[pk]) * d;
        if (u < (fminf(p1[pk], p2[pk]) - p0[pk] - (float)th * kThr) * t || u > (fmaxf(p1[pk], p2[pk]) - p0[pk] + (float)th * kThr) * t) return 0;
        t = d < 0.0f ? -t : +t;
        d = fabsf(d);
        w = (p1[qk] - p0[qk]) * t;
        if (w < (fminf(q1[qk], q2[qk]) - p0[qk] - (float)th * kThr) * d || w > (fmaxf(q1[qk], q2[qk]) - p0[qk] + (float)th * kThr) * d) return 0;
    }
    else
    {
        float ep[3], eq[3], pr[3], qr[3], mx[3], nx[3];
        for (int a = 0; a < 3; a++) { ep[a] = p2[a] - p1[a]; eq[a] = q2[a] - q1[a]; pr[a] = p1[a] - p0[a]; qr[a] = q1[a] - p0[a]; }
        mx[0] = eq[1] * ep[2] - ep[1] * eq[2]; mx[1] = eq[2] * ep[0] - ep[2] * eq[0]; mx[2] = eq[0] * ep[1] - ep[0] * eq[1];
        nx[0] = qr[1] * pr[2] - pr[1] * qr[2]; nx[1] = qr[2] * pr[0] - pr[2] * qr[0]; nx[2] = qr[0] * pr[1] - pr[0] * qr[1];
        t = dot3(qr, mx);
        s = t < 0.0f ? -1.0f : +1.0f;
        t = fabsf(t);
        u = dot3(eq, nx) * s;
        if (u < (0.0f - (float)th * kThr) * t || u > (1.0f + (float)th * kThr) * t) return 0;
        t *= s;
        d = dot3(pr, mx);
        s = d < 0.0f ? -1.0f : +1.0f;
        d = fabsf(d);
        w = dot3(ep, nx) * s;
        if (w < (0.0f - (float)th * kThr) * d || w > (1.0f + (float)th * kThr) * d) return 0;
        t *= s;
    }
    return t > (1.0f + kThr) * d ? 1 : t >= (1.0f - kThr) * d ? 3 : t > (0.0f + kThr) * d ? 2 : t >= (0.0f - kThr) * d ? 4 : 0;
}

static inline float asin32(float a) { return a <= -1.0f ? -(float)(3.14159265358979323846 / 2.0) : a >= 1.0f ? (float)(3.14159265358979323846 / 2.0) : asinf(a); }
static inline float acos32(float a) { return a <= -1.0f ? (float)3.14159265358979323846 : a >= 1.0f ? 0.0f : acosf(a); }
static inline float len3(const float *a) { const float d = dot3(a, a); return d <= 0.0f ? 0.0f : sqrtf(d); }

/* bbox_shad 1004-1153: may the box `i1` cast a shadow on the box `i2` as seen from the point `pps` (a light) */
int QrSideGeom::shad(const float *pps, int i1, int i2) const
{
    const Box &n1 = box[i1], &n2 = box[i2];
    if (n1.rad == kInf || n2.rad == kInf || i1 == i2) return 1;
    /* clip relations between two surfaces, RT_OPTS_SHADOW_EXT2 */
    if (!n1.array && !n2.array && (surf_clip(i2, i1) != 0 || surf_clip(i1, i2) != 0)) return 1;

    /* cones around the bounding spheres */
    float v1[3], v2[3];
    for (int a = 0; a < 3; a++) { v1[a] = n1.mid[a] - pps[a]; v2[a] = n2.mid[a] - pps[a]; }
    const float l1 = len3(v1), l2 = len3(v2);
    float ang = dot3(v1, v2);
    ang = l1 <= kThr ? 0.0f : ang / l1;
    const float a1 = l1 >= n1.rad && l1 > kThr ? asin32(n1.rad / l1) : (float)(2.0 * 3.14159265358979323846);
    ang = l2 <= kThr ? 0.0f : ang / l2;
    const float a2 = l2 >= n2.rad && l2 > kThr ? asin32(n2.rad / l2) : (float)(2.0 * 3.14159265358979323846);
    ang = acos32(ang);
    if (a1 + a2 < ang) return 0;

    /* the caster's sphere entirely behind the receiver's */
    const float dv[3] = { n1.mid[0] - n2.mid[0], n1.mid[1] - n2.mid[1], n1.mid[2] - n2.mid[2] };
    if (n1.rad + n2.rad < len3(dv) && l1 > l2) return 0;

    /* box geometry, RT_OPTS_SHADOW_EXT1 */
    if (n1.nverts == 0 || n2.nverts == 0) return 1;
    if (node_bbox(n1, pps) != 0) return 1;
    for (int q = 0; q < n1.nverts; q++)
        for (int f = 0; f < n2.nfaces; f++)
            if (vert_face(pps, n1.verts[q], +1, n2.verts[kFaces[f][0]], n2.verts[kFaces[f][1]], n2.verts[kFaces[f][3]],
                          n2.face_k[f], n2.face_i[f], n2.face_j[f]) == 1) return 1;
    for (int q = 0; q < n2.nverts; q++)
        for (int f = 0; f < n1.nfaces; f++)
        {
            const int k = vert_face(pps, n2.verts[q], +1, n1.verts[kFaces[f][0]], n1.verts[kFaces[f][1]], n1.verts[kFaces[f][3]],
                                    n1.face_k[f], n1.face_i[f], n1.face_j[f]);
            if (k == 2 || k == 4) return 1;
        }
    for (int e = 0; e < n1.nedges; e++)
        for (int g = 0; g < n2.nedges; g++)
            if (edge_edge(pps, +1, n1.verts[kEdges[e][0]], n1.verts[kEdges[e][1]], n1.edge_k[e],
                          n2.verts[kEdges[g][0]], n2.verts[kEdges[g][1]], n2.edge_k[g]) == 1) return 1;
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
