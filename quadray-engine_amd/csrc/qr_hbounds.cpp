/*
 * qr_hbounds.cpp - bounding and clipping boxes of a hierarchy (include/qr_hierarchy.h: qr_hierarchy_bounds).
 *
 * What the reference computes in phase 1 / 2 of rt_Scene::render after the transform update:
 *   rt_Surface::update_minmax / recalc_minmax / direct_minmax / invert_minmax   core/engine/object.cpp:2534-2799
 *   adjust_minmax of the plane and the nine quadric shapes                        2508-2527, 2957-2992, 3138-3956
 *   rt_Surface::update_bounds, rt_Node::update_bbgeom                             2801-2845, 849-1091
 *   rt_Array::update_bounds (inbox / trbox / bvbox, the records of the bounding volumes)   1830-2318
 * restated over the flat node table.  fp32, the reference's operation order; RT_INF is FLT_MAX and is tested with ==
 * like there.  Results are compared bit for bit with the engine's (tests/test_hierarchy.py).  Built with -ffp-contract=off.
 *
 * Structure (ours): every shape is a small table of per-axis limits ("how far the shape reaches along a local axis given
 * what the other axes are cut to"), and ONE routine applies a table to a source box -- the reference writes the same
 * comparisons out per shape.
 */
#include "qr_internal.h"
#include "qr_hierarchy.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include <string>

namespace {

const float kInf = FLT_MAX;

inline float mx(float a, float b) { return a > b ? a : b; }        /* RT_MAX / RT_MIN: the second operand wins ties and NaNs */
inline float mn(float a, float b) { return a < b ? a : b; }
inline float rsqrt0(float a) { return a <= 0.0f ? 0.0f : sqrtf(a); }   /* RT_SQRT */

struct V3f { float v[3]; float &operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

/* how a limit takes part: in the clip box a side that lies at or beyond the limit is no clipper at all (-> infinity);
 * in the bounding box the side is pulled in to the limit */
enum LimMode
{
    L_NONE = 0,         /* the shape does not end along this axis                                                   */
    L_CLOSED,           /* cbox: side <= limit drops out; bbox: always pulled in                                    */
    L_OPEN_SELF,        /* cbox: side <  limit drops out; bbox: pulled in only when the surface adjusts its own box */
    L_CLOSED_SELF       /* cbox: side <= limit drops out; bbox: pulled in only in self-adjust                       */
};
struct Lim { float at; LimMode mode; };
struct Reach { Lim lo[3], hi[3]; bool flat; };     /* flat: a plane -- no extent along K at all */

struct SurfIn
{
    int tag; const float *shape;        /* qr_node::tag / shape */
    const float *lmin, *lmax;           /* the axis clippers of the scene description, local axes I, J, K */
    int map[3], sgn[3];                 /* local axis a is sub-world axis map[a], with sign sgn[a] */
    float scl[3];                       /* scalers by sub-world axis */
    float pps[3];                       /* position in the sub-world space; zero for a surface that is its own transform node */
};

/* the limits of a shape for the source box [smin, smax] (local axes) */
Reach shape_reach(const SurfIn &s, const float *smin, const float *smax)
{
    Reach r;
    for (int a = 0; a < 3; a++) { r.lo[a] = { 0.0f, L_NONE }; r.hi[a] = { 0.0f, L_NONE }; }
    r.flat = s.tag == 0;
    const float p0 = s.shape[0], p1 = s.shape[1];
    auto sym = [&](int a, float reach, LimMode m) { r.lo[a] = { -reach, m }; r.hi[a] = { +reach, m }; };
    auto widest = [&](int a) { return mx(fabsf(smin[a]), fabsf(smax[a])); };
    switch (s.tag)
    {
    case 1:     /* cylinder, 3138-3176: radius across */
        sym(0, fabsf(p0), L_CLOSED); sym(1, fabsf(p0), L_CLOSED);
        break;
    case 2:     /* sphere, 3232-3284: along every axis the radius of the widest circle the two other axes' cuts leave */
    {
        float rad[3] = { fabsf(p0), fabsf(p0), fabsf(p0) };
        for (int k = 0; k < 3; k++)
        {
            const float top = smin[k] > 0.0f ? +smin[k] : smax[k] < 0.0f ? -smax[k] : 0.0f;
            const float rr = rsqrt0(mx(p0 * p0 - top * top, 0.0f));
            const int i = (k + 1) % 3, j = (k + 2) % 3;
            if (rad[i] > rr) rad[i] = rr;
            if (rad[j] > rr) rad[j] = rr;
        }
        for (int a = 0; a < 3; a++) sym(a, rad[a], L_CLOSED);
        break;
    }
    case 3:     /* cone, 3353-3393 */
    case 5:     /* hyperboloid, 3581-3621: the same with the waist term */
    {
        const float rat = fabsf(p0), hyp = s.tag == 5 ? p1 : 0.0f;
        float top = widest(2);
        const float rad = top != kInf ? (s.tag == 5 ? rsqrt0(top * top * rat * rat + hyp) : top * rat) : kInf;
        const float mi = widest(0), mj = widest(1);
        top = mn(mi != kInf && mj != kInf ? (s.tag == 5 ? rsqrt0(mi * mi + mj * mj - hyp) : rsqrt0(mi * mi + mj * mj)) / rat : top, top);
        sym(0, rad, L_CLOSED); sym(1, rad, L_CLOSED); sym(2, top, L_OPEN_SELF);
        break;
    }
    case 4:     /* paraboloid, 3462-3513 */
    case 6:     /* paracylinder, 3692-3740: the same without axis J */
    {
        const float par = p0;
        float top = mx(par < 0.0f ? -smin[2] : +smax[2], 0.0f);
        const float rad = top != kInf ? rsqrt0(top * fabsf(par)) : kInf;
        const float mi = widest(0);
        if (s.tag == 4)
        {
            const float mj = widest(1);
            top = mn(mi != kInf && mj != kInf ? (mi * mi + mj * mj) / fabsf(par) : top, top);
            sym(1, rad, L_CLOSED);
        }
        else top = mn(mi != kInf ? (mi * mi) / fabsf(par) : top, top);
        sym(0, rad, L_CLOSED);
        /* the vertex side ends at 0 whatever the clippers, the open side at `top` */
        if (par > 0.0f) { r.lo[2] = { 0.0f, L_CLOSED }; r.hi[2] = { +top, L_OPEN_SELF }; }
        else if (par < 0.0f) { r.lo[2] = { -top, L_OPEN_SELF }; r.hi[2] = { 0.0f, L_CLOSED }; }
        else { r.lo[2] = { -top, L_NONE }; r.hi[2] = { +top, L_NONE }; }
        break;
    }
    case 7:     /* hypercylinder, 3809-3848 */
    {
        const float rat = fabsf(p0), hyp = p1;
        float top = widest(2);
        const float rad = top != kInf ? rsqrt0(top * top * rat * rat + hyp) : kInf;
        const float mi = widest(0);
        top = mn(mi != kInf ? rsqrt0(mi * mi - hyp) / rat : top, top);
        sym(0, rad, L_CLOSED); sym(2, top, L_OPEN_SELF);
        break;
    }
    case 8:     /* hyperparaboloid, 3916-3944 */
    {
        const float rd1 = mx(-smin[0], +smax[0]), rd2 = mx(-smin[1], +smax[1]);
        const float tp1 = rd1 * rd1 / fabsf(p0), tp2 = rd2 * rd2 / fabsf(p1);
        r.lo[2] = { -tp2, L_CLOSED_SELF }; r.hi[2] = { +tp1, L_CLOSED_SELF };
        break;
    }
    default: break;
    }
    return r;
}

/*
 * adjust_minmax: source box (local) -> bounding box and / or clipping box (local).  `self`: the surface adjusts its own
 * boxes (the reference's cb: a clip box was passed).  Either destination may be null.
 */
void adjust(const SurfIn &s, const float *smin, const float *smax, float *bmin, float *bmax, float *cmin, float *cmax)
{
    const bool self = cmin != nullptr;
    if (cmin != nullptr)
        for (int a = 0; a < 3; a++)
        {
            /* rt_Surface::adjust_minmax 2508-2527: a side inside the scene description's own clipper is not a clipper */
            cmin[a] = smin[a] > s.lmin[a] ? -kInf : smin[a];
            cmax[a] = smax[a] < s.lmax[a] ? +kInf : smax[a];
        }
    const Reach r = shape_reach(s, smin, smax);
    if (cmin != nullptr)
    {
        if (r.flat) { cmin[2] = -kInf; cmax[2] = +kInf; }
        for (int a = 0; a < 3; a++)
        {
            const Lim &lo = r.lo[a], &hi = r.hi[a];
            if (lo.mode == L_CLOSED || lo.mode == L_CLOSED_SELF) { if (cmin[a] <= lo.at) cmin[a] = -kInf; }
            else if (lo.mode == L_OPEN_SELF) { if (cmin[a] < lo.at) cmin[a] = -kInf; }
            if (hi.mode == L_CLOSED || hi.mode == L_CLOSED_SELF) { if (cmax[a] >= hi.at) cmax[a] = +kInf; }
            else if (hi.mode == L_OPEN_SELF) { if (cmax[a] > hi.at) cmax[a] = +kInf; }
        }
    }
    if (bmin != nullptr)
    {
        for (int a = 0; a < 3; a++)
        {
            const Lim &lo = r.lo[a], &hi = r.hi[a];
            const bool pull_lo = lo.mode == L_CLOSED || ((lo.mode == L_OPEN_SELF || lo.mode == L_CLOSED_SELF) && self);
            const bool pull_hi = hi.mode == L_CLOSED || ((hi.mode == L_OPEN_SELF || hi.mode == L_CLOSED_SELF) && self);
            bmin[a] = pull_lo ? mx(smin[a], lo.at) : smin[a];
            bmax[a] = pull_hi ? mn(smax[a], hi.at) : smax[a];
        }
        if (r.flat) { bmin[2] = 0.0f; bmax[2] = 0.0f; }
    }
}

/* direct_minmax 2575-2605: local box -> sub-world box (axis map, scalers, position); in place allowed */
void to_subworld(const SurfIn &s, const float *smin, const float *smax, float *dmin, float *dmax)
{
    float tmin[3], tmax[3];
    for (int a = 0; a < 3; a++)
    {
        tmin[s.map[a]] = s.sgn[a] > 0 ? +smin[a] : -smax[a];
        tmax[s.map[a]] = s.sgn[a] > 0 ? +smax[a] : -smin[a];
    }
    for (int x = 0; x < 3; x++)
    {
        tmin[x] = tmin[x] == -kInf ? -kInf : tmin[x] * s.scl[x];
        tmax[x] = tmax[x] == +kInf ? +kInf : tmax[x] * s.scl[x];
    }
    for (int x = 0; x < 3; x++)
    {
        dmin[x] = tmin[x] == -kInf ? -kInf : tmin[x] + s.pps[x];
        dmax[x] = tmax[x] == +kInf ? +kInf : tmax[x] + s.pps[x];
    }
}

/* invert_minmax 2534-2566: sub-world box -> local box */
void to_local(const SurfIn &s, const float *smin, const float *smax, float *dmin, float *dmax)
{
    float tmin[3], tmax[3];
    for (int x = 0; x < 3; x++)
    {
        tmin[x] = smin[x] == -kInf ? -kInf : smin[x] - s.pps[x];
        tmax[x] = smax[x] == +kInf ? +kInf : smax[x] - s.pps[x];
    }
    for (int x = 0; x < 3; x++)
    {
        tmin[x] = tmin[x] == -kInf ? -kInf : tmin[x] / s.scl[x];
        tmax[x] = tmax[x] == +kInf ? +kInf : tmax[x] / s.scl[x];
    }
    for (int a = 0; a < 3; a++)
    {
        dmin[a] = s.sgn[a] > 0 ? +tmin[s.map[a]] : -tmax[s.map[a]];
        dmax[a] = s.sgn[a] > 0 ? +tmax[s.map[a]] : -tmin[s.map[a]];
    }
}

/* recalc_minmax 2611-2685, its three uses */

/* (1) from the scene description's own clippers: bounding box and, when asked for, clipping box, both in sub-world space */
void box_from_description(const SurfIn &s, float *bmin, float *bmax, float *cmin, float *cmax)
{
    adjust(s, s.lmin, s.lmax, bmin, bmax, cmin, cmax);
    to_subworld(s, bmin, bmax, bmin, bmax);
    if (cmin != nullptr) to_subworld(s, cmin, cmax, cmin, cmax);
}

/* (2) what an outer clipper `c` cuts off the sub-world box [smin, smax] of another surface: accumulated into [pmin, pmax] */
void cut_by_clipper(const SurfIn &c, const float *smin, const float *smax, float *pmin, float *pmax)
{
    float tmin[3], tmax[3], lmin[3], lmax[3];
    to_local(c, smin, smax, tmin, tmax);
    adjust(c, tmin, tmax, lmin, lmax, nullptr, nullptr);
    for (int a = 0; a < 3; a++)
    {
        tmin[a] = tmin[a] == lmin[a] ? -kInf : lmin[a];
        tmax[a] = tmax[a] == lmax[a] ? +kInf : lmax[a];
    }
    to_subworld(c, tmin, tmax, tmin, tmax);
    for (int x = 0; x < 3; x++) { pmin[x] = mx(pmin[x], tmin[x]); pmax[x] = mn(pmax[x], tmax[x]); }
}

/* (3) the accumulated cuts [smin, smax] applied: final bounding and clipping box */
void box_from_cuts(const SurfIn &s, const float *smin, const float *smax, float *bmin, float *bmax, float *cmin, float *cmax)
{
    float tmin[3], tmax[3];
    to_local(s, smin, smax, tmin, tmax);
    for (int a = 0; a < 3; a++) { tmin[a] = mx(tmin[a], s.lmin[a]); tmax[a] = mn(tmax[a], s.lmax[a]); }
    adjust(s, tmin, tmax, bmin, bmax, cmin, cmax);
    to_subworld(s, bmin, bmax, bmin, bmax);
    to_subworld(s, cmin, cmax, cmin, cmax);
}

/* one of the three boxes of an array, or the bounding box of a surface */
struct Box
{
    float bmin[3], bmax[3];
    float rad;                  /* while boxes are united: 0 empty, RT_INF unbounded, anything else finite; then the radius */
    float mid[3];
    float verts[8][3];
    int nverts;
};

void box_reset(Box &b) { for (int a = 0; a < 3; a++) { b.bmin[a] = +kInf; b.bmax[a] = -kInf; } b.rad = 0.0f; }

/* update_bbgeom 849-1091: corners (through `mtx` when the box lives in a transform node's space), centre, radius.
 * map: the owner's axis map (corner order follows local axes). */
void box_corners(Box &b, const int *map, const float (*mtx)[4], bool plane)
{
    static const int hi_i[8] = {1, 0, 0, 1, 1, 0, 0, 1}, hi_j[8] = {1, 1, 0, 0, 1, 1, 0, 0}, hi_k[8] = {1, 1, 1, 1, 0, 0, 0, 0};
    const int nv = plane ? 4 : 8;
    for (int q = 0; q < nv; q++)
    {
        float c[4];
        c[map[0]] = hi_i[q] ? b.bmax[map[0]] : b.bmin[map[0]];
        c[map[1]] = hi_j[q] ? b.bmax[map[1]] : b.bmin[map[1]];
        c[map[2]] = hi_k[q] ? b.bmax[map[2]] : b.bmin[map[2]];
        c[3] = 1.0f;
        if (mtx != nullptr)
            for (int x = 0; x < 3; x++)     /* matrix_mul_vector, rtgeom.cpp:59-77: four products summed left to right */
                b.verts[q][x] = mtx[0][x] * c[0] + mtx[1][x] * c[1] + mtx[2][x] * c[2] + mtx[3][x] * c[3];
        else
            for (int x = 0; x < 3; x++) b.verts[q][x] = c[x];
    }
    b.nverts = nv;
    const float f = 1.0f / (float)nv;
    b.mid[0] = b.mid[1] = b.mid[2] = 0.0f;
    for (int q = 0; q < nv; q++) for (int x = 0; x < 3; x++) b.mid[x] += b.verts[q][x] * f;
    float far2 = 0.0f;
    for (int q = 0; q < nv; q++)
    {
        const float d0 = b.mid[0] - b.verts[q][0], d1 = b.mid[1] - b.verts[q][1], d2 = b.mid[2] - b.verts[q][2];
        const float dd = d0 * d0 + d1 * d1 + d2 * d2;
        if (far2 < dd) far2 = dd;
    }
    b.rad = rsqrt0(far2);
}

/* the two ways a member's box enters an enclosing box (1934-2010, 2145-2180) */
void unite_minmax(Box &dst, const Box &src)
{
    if (src.rad != kInf)
        for (int x = 0; x < 3; x++)
        {
            if (dst.bmin[x] > src.bmin[x]) dst.bmin[x] = src.bmin[x];
            if (dst.bmax[x] < src.bmax[x]) dst.bmax[x] = src.bmax[x];
        }
    if (dst.rad < src.rad) dst.rad = src.rad;
}

void unite_corners(Box &dst, const Box &src)
{
    if (src.rad != kInf)
        for (int q = 0; q < src.nverts; q++)
            for (int x = 0; x < 3; x++)
            {
                if (dst.bmin[x] > src.verts[q][x]) dst.bmin[x] = src.verts[q][x];
                if (dst.bmax[x] < src.verts[q][x]) dst.bmax[x] = src.verts[q][x];
            }
    if (dst.rad < src.rad) dst.rad = src.rad;
}

struct Arr { Box in, bv, tr; };

} /* namespace */

extern "C" int qr_hierarchy_bounds(const void *blob, uint64_t size, const qr_node *nodes, int32_t n, uint32_t opts,
                                   qr_node_bounds *out)
{
    if (blob == nullptr || nodes == nullptr || out == nullptr || n <= 0) return qr_fail(QR_ERR_ARG, "null argument");
    qr_scene_view v;
    if (qr_scene_view_init(&v, blob, size) != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot");
    std::vector<qr_node_state> st((size_t)n);
    int rc = qr_hierarchy_update(nodes, n, opts, st.data());
    if (rc != QR_OK) return rc;
    const int n_srf = (int)v.hdr->n_srf, n_elm = (int)v.hdr->n_elm;

    /* snapshot record -> node */
    std::vector<int> node_of((size_t)n_srf, -1);
    for (int i = 0; i < n; i++)
    {
        const qr_node &nd = nodes[i];
        const bool owns = (nd.tag >= 0 && nd.tag < QR_TAG_SURFACE_MAX) || nd.tag == QR_NODE_ARRAY;
        if (!owns) continue;
        if (nd.srf >= n_srf || nd.inb >= n_srf || nd.bvb >= n_srf) return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": record index outside the snapshot");
        if (nd.srf >= 0) node_of[(size_t)nd.srf] = i;
        if (nd.bvnode >= i || (nd.bvnode >= 0 && nodes[nd.bvnode].tag != QR_NODE_ARRAY))
            return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": its bounding-volume node must be an array in front of it");
    }
    auto is_srf = [&](int i) { return nodes[i].tag >= 0 && nodes[i].tag < QR_TAG_SURFACE_MAX; };
    auto surf_in = [&](int i) {
        SurfIn s;
        const qr_node &nd = nodes[i]; const qr_node_state &q = st[(size_t)i];
        s.tag = nd.tag; s.shape = nd.shape; s.lmin = nd.lmin; s.lmax = nd.lmax;
        for (int a = 0; a < 3; a++) { s.map[a] = q.map[a]; s.sgn[a] = q.sgn[a]; s.scl[a] = q.scl[a]; }
        for (int x = 0; x < 3; x++) s.pps[x] = q.trnode == i ? 0.0f : q.mtx[12 + x];
        return s;
    };

    memset(out, 0, sizeof(qr_node_bounds) * (size_t)n);
    std::vector<Box> sbox((size_t)n);
    std::vector<Arr> abox((size_t)n);

    /* ---- surfaces: update_minmax 2690-2799, update_bounds 2801-2845 ---- */
    /* bounding boxes from the descriptions first: a clipper's cut needs the clipped surface's, not its own */
    for (int i = 0; i < n; i++)
    {
        if (!is_srf(i)) continue;
        const SurfIn s = surf_in(i);
        qr_node_bounds &o = out[i];
        box_from_description(s, o.bmin, o.bmax, o.cmin, o.cmax);
    }
    if (opts & QR_OPTS_ADJUST)
        for (int i = 0; i < n; i++)
        {
            if (!is_srf(i) || nodes[i].srf < 0 || st[(size_t)i].trnode == i) continue;
            const int head = v.srf[nodes[i].srf].clip;
            if (head == QR_NULL) continue;
            const SurfIn s = surf_in(i);
            qr_node_bounds &o = out[i];
            float bmin[3], bmax[3];
            box_from_description(s, bmin, bmax, nullptr, nullptr);
            float cutmin[3] = { -kInf, -kInf, -kInf }, cutmax[3] = { +kInf, +kInf, +kInf };
            bool in_accum = false;
            int guard = 0;
            for (int e = head; e != QR_NULL; e = v.elm[e].next)
            {
                if (e < 0 || e >= n_elm || ++guard > n_elm) return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": broken clipper list");
                const qr_elem &el = v.elm[e];
                if (el.simd == QR_NULL) { in_accum = !in_accum; continue; }         /* accumulator segment: its members are holes */
                if (in_accum || el.simd < 0 || el.simd >= n_srf) continue;
                const int c = node_of[(size_t)el.simd];
                if (c < 0 || !is_srf(c) || nodes[c].tag == 0) continue;             /* arrays (trnode groups) and planes cut nothing */
                if (st[(size_t)c].trnode != st[(size_t)i].trnode || el.data != 1) continue;   /* same space, RT_REL_MINUS_OUTER only */
                cut_by_clipper(surf_in(c), bmin, bmax, cutmin, cutmax);
            }
            box_from_cuts(s, cutmin, cutmax, o.bmin, o.bmax, o.cmin, o.cmax);
        }
    for (int i = 0; i < n; i++)
    {
        if (!is_srf(i)) continue;
        const qr_node_state &q = st[(size_t)i];
        qr_node_bounds &o = out[i];
        Box &b = sbox[(size_t)i];
        for (int x = 0; x < 3; x++) { b.bmin[x] = o.bmin[x]; b.bmax[x] = o.bmax[x]; }
        b.rad = kInf; b.nverts = 0;                                                 /* rt_Surface's constructor, 2365 */
        if (nodes[i].nverts != 0)
        {
            const int map[3] = { q.map[0], q.map[1], q.map[2] };
            box_corners(b, map, q.trnode >= 0 ? (const float (*)[4])st[(size_t)q.trnode].mtx : nullptr, nodes[i].tag == 0);
        }
        for (int x = 0; x < 3; x++) o.mid[x] = b.mid[x];
        o.rad = b.rad; o.nverts = b.nverts;
    }

    /* ---- arrays: rt_Array::update_bounds.  The reference recurses; children follow their parents in the table, so a sweep
     *      from the back meets every node after all of its descendants, and a parent's boxes are reset before any
     *      descendant contributes because contributions are made in a second, forward pass per finished child ---- */
    for (int i = 0; i < n; i++) if (nodes[i].tag == QR_NODE_ARRAY) { box_reset(abox[(size_t)i].in); box_reset(abox[(size_t)i].bv); box_reset(abox[(size_t)i].tr); }
    /* post-order: a node contributes once its own subtree is finished */
    std::vector<int> last_desc((size_t)n);
    for (int i = n - 1; i >= 0; i--)
    {
        last_desc[(size_t)i] = i;
    }
    for (int i = n - 1; i > 0; i--) { const int p = nodes[i].parent; if (p >= 0 && last_desc[(size_t)p] < last_desc[(size_t)i]) last_desc[(size_t)p] = last_desc[(size_t)i]; }
    /* order of finishing: children in table order, each after its subtree = ascending by (last descendant, -index) */
    std::vector<int> order;
    order.reserve((size_t)n);
    {
        /* explicit depth-first walk in table order */
        std::vector<std::vector<int>> kids((size_t)n);
        int root = -1;
        for (int i = 0; i < n; i++) { if (nodes[i].parent >= 0) kids[(size_t)nodes[i].parent].push_back(i); else if (root < 0) root = i; }
        struct Fr { int node; size_t next; };
        std::vector<Fr> stk;
        for (int i = 0; i < n; i++)
        {
            if (nodes[i].parent >= 0) continue;
            stk.push_back({ i, 0 });
            while (!stk.empty())
            {
                Fr &f = stk.back();
                if (f.next < kids[(size_t)f.node].size()) { const int c = kids[(size_t)f.node][f.next++]; stk.push_back({ c, 0 }); }
                else { order.push_back(f.node); stk.pop_back(); }
            }
        }
    }
    auto finish_array = [&](int i) {
        /* 2181-2316: the array's own boxes once every member has contributed */
        const qr_node_state &q = st[(size_t)i];
        Arr &A = abox[(size_t)i];
        qr_node_bounds &o = out[i];
        const int map[3] = { q.map[0], q.map[1], q.map[2] };
        const float (*tm)[4] = q.trnode >= 0 ? (const float (*)[4])st[(size_t)q.trnode].mtx : nullptr;
        o.inb_form = 0; o.bvb_form = 0;
        if (A.in.rad != 0.0f && A.in.rad != kInf)
        {
            box_corners(A.in, map, tm, false);
            o.inb_form = 1;                             /* ellipsoid around the box */
            if (q.trnode == i && A.tr.rad != 0.0f) unite_minmax(A.tr, A.in);
            else if (q.trnode == i && A.bv.rad != 0.0f) unite_corners(A.bv, A.in);
            else if (q.trnode == i) o.inb_form = 2;     /* sphere around the box's centre, in the array's own space */
        }
        if (A.bv.rad != 0.0f && A.bv.rad != kInf)
        {
            box_corners(A.bv, map, nullptr, false);     /* always world space */
            o.bvb_form = 1;
        }
        if (A.tr.rad != 0.0f && A.tr.rad != kInf) box_corners(A.tr, map, tm, false);
        for (int x = 0; x < 3; x++)
        {
            o.inmin[x] = A.in.bmin[x]; o.inmax[x] = A.in.bmax[x]; o.bmin[x] = A.bv.bmin[x]; o.bmax[x] = A.bv.bmax[x];
            o.trmin[x] = A.tr.bmin[x]; o.trmax[x] = A.tr.bmax[x]; o.inmid[x] = A.in.mid[x]; o.mid[x] = A.bv.mid[x];
        }
        o.inrad = A.in.rad; o.rad = A.bv.rad; o.trrad = A.tr.rad;
    };
    for (int i : order)
    {
        const qr_node &nd = nodes[i];
        const bool arr = nd.tag == QR_NODE_ARRAY, srf = is_srf(i);
        if (arr) finish_array(i);
        if (!arr && !srf) continue;
        if (nd.parent < 0) continue;                    /* the root contributes to nobody */
        const qr_node_state &q = st[(size_t)i];
        const Box &own_bv = srf ? sbox[(size_t)i] : abox[(size_t)i].bv;
        /* 1895-2012: to the transform node's trbox, or to the inbox of a bounding-volume node inside that transform node */
        const int tn = q.trnode != i ? q.trnode : -1;
        if (tn >= 0)
        {
            const Box &src = srf ? sbox[(size_t)i] : abox[(size_t)i].in;
            Box &dst = (nd.bvnode >= 0 && st[(size_t)nd.bvnode].trnode == tn) ? abox[(size_t)nd.bvnode].in : abox[(size_t)tn].tr;
            if (src.rad != 0.0f) unite_minmax(dst, src);
        }
        /* 2014-2180: to the bounding-volume node's bvbox */
        const int bn = nd.bvnode;
        if (bn >= 0)
        {
            const bool inside_same_trnode = srf && q.trnode == st[(size_t)bn].trnode && q.trnode != i && q.trnode >= 0;
            const Box *src = inside_same_trnode ? nullptr : &own_bv;
            Box &dst = abox[(size_t)bn].bv;
            if (src != nullptr && src->rad != 0.0f && (q.trnode < 0 || arr)) unite_minmax(dst, *src);
            if (arr)
            {
                const Arr &A = abox[(size_t)i];
                src = (q.trnode != i || A.tr.rad != 0.0f || A.bv.rad == 0.0f) ? &A.in : nullptr;
            }
            if (src != nullptr && src->rad != 0.0f && q.trnode >= 0 && q.trnode != st[(size_t)bn].trnode) unite_corners(dst, *src);
        }
    }
    return QR_OK;
}
