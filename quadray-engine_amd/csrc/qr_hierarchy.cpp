/*
 * qr_hierarchy.cpp - animators and the hierarchical transform update (include/qr_hierarchy.h).
 *
 * Restates, over a flat parent-before-child node table, what the reference spreads over virtual
 * update_object / update_matrix / update_fields of its object classes (core/engine/object.cpp, lines cited at
 * each step).  fp32, one rounding per operation, the reference's operand order: the results are compared bit for
 * bit with the engine's (tests/test_hierarchy.py).  Built with -ffp-contract=off.
 */
#include "qr_internal.h"
#include "qr_hierarchy.h"

#include <cfloat>
#include <cmath>
#include <vector>
#include <algorithm>
#include <utility>
#include <cstdlib>
#include <cstring>

namespace {

enum { F_SCL = 1, F_ROT = 2, F_OBJ = 4 };       /* RT_UPDATE_FLAG_* (object.h:74-83) */

typedef float M4[4][4];

const M4 kIden = { {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1} };

/* row r of the product is `a` applied to row r of `b` (matrix_mul_vector / matrix_mul_matrix, rtgeom.cpp:59-97):
 * four products summed left to right */
void mul(M4 out, const M4 a, const M4 b)
{
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            out[r][c] = a[0][c] * b[r][0] + a[1][c] * b[r][1] + a[2][c] * b[r][2] + a[3][c] * b[r][3];
}

/* sine / cosine of an angle in degrees: exact at the multiples of 90 the engine treats as trivial, else the
 * single-precision library function of the double-precision radians (RT_SINA32 / RT_COSA32, rtbase.h:584-600) */
float sin_deg(float a)
{
    if (a == -270.0f || a == 90.0f) return 1.0f;
    if (a == -90.0f || a == 270.0f) return -1.0f;
    if (a == -180.0f || a == 0.0f || a == 180.0f) return 0.0f;
    return sinf((float)((double)a * 3.14159265358979323846 / 180.0));
}
float cos_deg(float a)
{
    if (a == -180.0f || a == 180.0f) return -1.0f;
    if (a == 0.0f) return 1.0f;
    if (a == -270.0f || a == -90.0f || a == 90.0f || a == 270.0f) return 0.0f;
    return cosf((float)((double)a * 3.14159265358979323846 / 180.0));
}

/* The object's matrix: scale first, then the turns about x, y and z, then the shift -- each step multiplied on from the
 * left with the full 4x4 product, so every element sees the roundings (and the signed zeros) the engine's chain of
 * matrix_mul_matrix calls produces (matrix_from_transform, rtgeom.cpp:102-162).  A turn about axis `ax` by angle w is
 * the identity with [p][p] = [q][q] = cos w, [p][q] = sin w, [q][p] = -sin w for p = ax + 1, q = ax + 2 (mod 3). */
void from_transform(M4 out, const qr_node &nd, bool with_scale)
{
    M4 acc, step, nxt;
    memcpy(acc, kIden, sizeof(M4));
    for (int ax = 0; ax < 3; ax++) if (with_scale) acc[ax][ax] = nd.scl[ax];
    for (int ax = 0; ax < 3; ax++)
    {
        const int p = (ax + 1) % 3, q = (ax + 2) % 3;
        const float sn = sin_deg(nd.rot[ax]), cs = cos_deg(nd.rot[ax]);
        memcpy(step, kIden, sizeof(M4));
        step[p][p] = cs; step[p][q] = sn; step[q][p] = -sn; step[q][q] = cs;
        mul(nxt, step, acc);
        memcpy(acc, nxt, sizeof(M4));
    }
    memcpy(step, kIden, sizeof(M4));
    for (int ax = 0; ax < 3; ax++) step[3][ax] = nd.pos[ax];
    mul(out, step, acc);
}

/* Inverse of the upper-left 3x3 as adjugate / determinant (matrix_inverse, rtgeom.cpp:167-193).  With indices taken
 * mod 3, element [r][c] of the adjugate is m[c+1][r+1] m[c+2][r+2] - m[c+2][r+1] m[c+1][r+2]; the determinant expands
 * along column 0 with the adjugate's first row, summed left to right; every element is then scaled by 1 / det (one
 * rounding each: the engine multiplies by the reciprocal, it does not divide). */
void inverse3(M4 out, const M4 m)
{
    memset(out, 0, sizeof(M4));
    float adj[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
        {
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
            adj[r][c] = m[c1][r1] * m[c2][r2] - m[c2][r1] * m[c1][r2];
        }
    float det = m[0][0] * adj[0][0];
    for (int k = 1; k < 3; k++) det = det + m[k][0] * adj[0][k];
    const float rdet = 1.0f / det;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) out[r][c] = adj[r][c] * rdet;
}

/* what a node hands to its children (rt_Array::update_object, object.cpp:1739-1756) */
struct Pass { int flags; int trnode; M4 mtx; };

bool all_in(const float v[3], const float *set, int n)
{
    int hits = 0;
    for (int k = 0; k < n; k++) for (int a = 0; a < 3; a++) if (v[a] == set[k]) hits++;
    return hits == 3;
}

void set_identity_map(qr_node_state &s, float sx, float sy, float sz)
{
    for (int i = 0; i < 4; i++) { s.map[i] = i; s.sgn[i] = 1; }
    s.scl[0] = sx; s.scl[1] = sy; s.scl[2] = sz; s.scl[3] = 1.0f;
}

/* update_status + update_matrix of one object (object.cpp:175-389); `in` is what the parent passed down */
void update_node(const qr_node &nd, int self, const Pass &in, const qr_node_state *all, uint32_t opts, qr_node_state &s)
{
    static const float unit_scl[] = { -1.0f, 1.0f };
    static const float right_rot[] = { -270.0f, -180.0f, -90.0f, 0.0f, 90.0f, 180.0f, 270.0f };
    M4 &mtx = *(M4 *)s.mtx;
    M4 own, tmp;

    /* update_status 204-213: transform flags and trnode come from the hierarchy (every call is a full update) */
    s.obj_has_trm = in.flags & (F_SCL | F_ROT);
    s.trnode = in.trnode;
    set_identity_map(s, 1.0f, 1.0f, 1.0f);      /* the engine leaves them as they were; a trivial matrix sets all of them below */

    /* 236-262: is the object's own transform trivial (unit scalers, right-angle rotation)? */
    s.mtx_has_trm = 0;
    if (!all_in(nd.scl, unit_scl, 2)) s.mtx_has_trm |= F_SCL;
    if (!all_in(nd.rot, right_rot, 7)) s.mtx_has_trm |= F_ROT;

    if (!(s.mtx_has_trm & F_ROT))
    {
        /* 265-304: no rotation of its own: the matrix stays an axis permutation with signs and scalers */
        from_transform(own, nd, true);
        mul(mtx, in.mtx, own);
        if (s.obj_has_trm == F_SCL) { s.mtx_has_trm = s.obj_has_trm; s.obj_has_trm = 0; }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
            {
                bool same = true;       /* row i has its non-zeros where row j of the identity has them */
                for (int c = 0; c < 3; c++) same = same && ((mtx[i][c] != 0.0f) == (c == j));
                if (same)
                {
                    s.map[i] = j;
                    s.sgn[i] = mtx[i][j] < 0.0f ? -1 : mtx[i][j] > 0.0f ? 1 : 0;
                    s.scl[j] = fabsf(mtx[i][j]);
                }
            }
        s.map[3] = 3; s.sgn[3] = 1; s.scl[3] = 1.0f;
    }
    else
    {
        /* 307-341: a rotation of its own makes the object a transform node; scalers stay out of the matrix,
         * the solvers apply them; under another trnode the matrix is taken to world space first */
        from_transform(own, nd, false);
        if (s.trnode < 0) mul(mtx, in.mtx, own);
        else
        {
            mul(tmp, *(const M4 *)all[s.trnode].mtx, in.mtx);
            mul(mtx, tmp, own);
        }
        s.trnode = self;
        s.obj_has_trm |= F_ROT;
        set_identity_map(s, nd.scl[0], nd.scl[1], nd.scl[2]);
    }

    /* 343-350: without the FSCALE optimisation a rotated object is treated as scaled too */
    if ((s.obj_has_trm & F_ROT) && !(opts & QR_OPTS_FSCALE)) s.obj_has_trm |= F_SCL;

    /* 352-381: cameras and lights (and, without transform caching, everything) do not stay relative to a trnode:
     * their matrix goes to world space and they become their own */
    if (s.trnode >= 0 && s.trnode != self && (!(opts & QR_OPTS_TARRAY) || nd.tag > QR_TAG_SURFACE_MAX))
    {
        mul(tmp, *(const M4 *)all[s.trnode].mtx, mtx);
        memcpy(mtx, tmp, sizeof(M4));
        s.trnode = self;
        s.obj_has_trm |= s.mtx_has_trm;
        set_identity_map(s, 1.0f, 1.0f, 1.0f);
    }
}

int run_update(const qr_node *nodes, int32_t n, uint32_t opts, qr_node_state *out, std::string &err)
{
    if (nodes == nullptr || out == nullptr || n <= 0) { err = "bad node table"; return QR_ERR_ARG; }
    std::vector<Pass> pass((size_t)n);
    for (int i = 0; i < n; i++)
    {
        const qr_node &nd = nodes[i];
        if (nd.parent >= i || nd.parent < -1) { err = "node " + std::to_string(i) + ": parents must precede children"; return QR_ERR_ARG; }
        if (nd.parent >= 0 && nodes[nd.parent].tag != QR_NODE_ARRAY) { err = "node " + std::to_string(i) + ": parent is not an array"; return QR_ERR_ARG; }
        Pass root; root.flags = 0; root.trnode = -1; memcpy(root.mtx, kIden, sizeof(M4));
        const Pass &in = nd.parent < 0 ? root : pass[(size_t)nd.parent];
        memset(&out[i], 0, sizeof(qr_node_state));
        update_node(nd, i, in, out, opts, out[i]);
        if (nd.tag == QR_NODE_ARRAY)
        {
            /* rt_Array::update_matrix 1708-1733 + update_object 1750-1755: children get the array's matrix, or, under
             * an array that is a transform node, only its scalers; and its flags on top of the inherited ones */
            Pass &p = pass[(size_t)i];
            p.flags = in.flags | out[i].mtx_has_trm | F_OBJ;
            p.trnode = out[i].trnode;
            if (out[i].trnode == i)
            {
                memcpy(p.mtx, kIden, sizeof(M4));
                p.mtx[0][0] = out[i].scl[0]; p.mtx[1][1] = out[i].scl[1]; p.mtx[2][2] = out[i].scl[2];
            }
            else memcpy(p.mtx, out[i].mtx, sizeof(M4));
        }
    }
    return QR_OK;
}

bool is_surface(int tag) { return tag >= 0 && tag < QR_TAG_SURFACE_MAX; }

/* a_map / a_sgn / trnode of a surface or array record (rt_Surface::update_fields 2486-2502, rt_Array 1770-1788) */
void put_mapping(qr_surface &r, const qr_node_state &s, const qr_node *nodes)
{
    r.has_trm = s.obj_has_trm;
    r.shift = s.trnode >= 0 ? 1 : 0;
    r.axes = (uint32_t)(s.map[0] & 3) | (uint32_t)(s.map[1] & 3) << 2 | (uint32_t)(s.map[2] & 3) << 4
           | (s.sgn[0] >= 0 ? 0u : 1u) << 8 | (s.sgn[1] >= 0 ? 0u : 1u) << 9 | (s.sgn[2] >= 0 ? 0u : 1u) << 10;
    r.trnode = s.trnode >= 0 ? nodes[s.trnode].srf : QR_NULL;
}

/* quadric coefficients: shape in local axes (update_fields of the nine shapes, 3120-3910), then the scalers
 * (rt_Quadric::commit_fields 3034-3063) */
void put_quadric(qr_surface &r, const qr_node &nd, const qr_node_state &s)
{
    float sci[4] = { 1.0f, 1.0f, 1.0f, 0.0f }, scj[3] = { 0.0f, 0.0f, 0.0f };
    const int mi = s.map[0], mj = s.map[1], mk = s.map[2];
    const float sk = (float)s.sgn[2];
    switch (nd.tag)
    {
    case 1: sci[mk] = 0.0f; sci[3] = nd.shape[0] * nd.shape[0]; break;                       /* cylinder        */
    case 2: sci[3] = nd.shape[0] * nd.shape[0]; break;                                       /* sphere          */
    case 3: sci[mk] = -(nd.shape[0] * nd.shape[0]); break;                                   /* cone            */
    case 4: sci[mk] = 0.0f; scj[mk] = nd.shape[0] * sk; break;                               /* paraboloid      */
    case 5: sci[mk] = -(nd.shape[0] * nd.shape[0]); sci[3] = nd.shape[1]; break;             /* hyperboloid     */
    case 6: sci[mj] = 0.0f; sci[mk] = 0.0f; scj[mk] = nd.shape[0] * sk; break;               /* paracylinder    */
    case 7: sci[mj] = 0.0f; sci[mk] = -(nd.shape[0] * nd.shape[0]); sci[3] = nd.shape[1]; break; /* hypercylinder */
    case 8: sci[mi] = 1.0f / +fabsf(nd.shape[0]); sci[mj] = 1.0f / -fabsf(nd.shape[1]);      /* hyperparaboloid */
            sci[mk] = 0.0f; scj[mk] = 1.0f * sk; break;
    default: return;
    }
    float isc[3];
    for (int a = 0; a < 3; a++) isc[a] = 1.0f / s.scl[a];
    for (int a = 0; a < 3; a++)
    {
        sci[a] *= isc[a] * isc[a];
        scj[a] *= isc[a];
    }
    for (int a = 0; a < 4; a++) r.sci[a] = sci[a];
    for (int a = 0; a < 3; a++) r.scj[a] = scj[a] * 0.5f;
}

bool same_bits(const void *a, const void *b, size_t n) { return memcmp(a, b, n) == 0; }

} /* namespace */

extern "C" int qr_hierarchy_update(const qr_node *nodes, int32_t n, uint32_t opts, qr_node_state *out)
{
    std::string err;
    const int rc = run_update(nodes, n, opts, out, err);
    return rc == QR_OK ? QR_OK : qr_fail(rc, err);
}

extern "C" int qr_hierarchy_animate(qr_node *nodes, int32_t n, int64_t time, int64_t *node_time,
                                    const qr_anim_fn *fns, void *const *users, int32_t n_fns)
{
    if (nodes == nullptr || node_time == nullptr || n <= 0) return qr_fail(QR_ERR_ARG, "bad node table");
    for (int i = 0; i < n; i++)
    {
        const int k = nodes[i].anim;
        if (k >= 0)
        {
            if (fns == nullptr || k >= n_fns || fns[k] == nullptr) return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": no animator in slot " + std::to_string(k));
            /* object.cpp:182-190: once per time value, the first call sees last_time 0 */
            if (node_time[i] != time) fns[k](time, node_time[i] < 0 ? 0 : node_time[i], nodes[i].scl, users ? users[k] : nullptr);
        }
        node_time[i] = time;
    }
    return QR_OK;
}

extern "C" void qr_anim_spin(int64_t time, int64_t last_time, float *trm, void *user)
{
    const qr_anim_params *p = (const qr_anim_params *)user;
    float *rot = trm + 3;
    const float t = (float)(time - last_time) / 50.0f;
    rot[p->axis] += t * p->rate;
    if (rot[p->axis] >= 360.0f) rot[p->axis] -= 360.0f;
}

extern "C" void qr_anim_swing(int64_t time, int64_t, float *trm, void *user)
{
    const qr_anim_params *p = (const qr_anim_params *)user;
    const float t = (float)time / p->period;
    trm[3 + p->axis] = (float)((double)p->rate * sin((double)t));
}

extern "C" int qr_hierarchy_apply(const void *blob, uint64_t size, const qr_node *base, const qr_node *next, int32_t n,
                                  uint32_t opts, int32_t camera, uint32_t flags, void **out_blob, uint64_t *out_size)
{
    if (blob == nullptr || next == nullptr || out_blob == nullptr || out_size == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    qr_scene_view v;
    if (qr_scene_view_init(&v, blob, size) != 0) return qr_fail(QR_ERR_ARG, "not a snapshot");
    std::string err;
    int rc = qr_snapshot_validate(v, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    std::vector<qr_node_state> st((size_t)(n > 0 ? n : 0)), st0;
    bool regroup = false;                                   /* surfaces change the array that is their transform node */
    std::vector<int> self_of;                               /* surfaces whose transform node changes (another array, themselves, none) */
    std::vector<int> plane_tex;                             /* planes whose axis scalers leave 1: texture scale / offset of their materials */
    rc = run_update(next, n, opts, st.data(), err);
    if (rc != QR_OK) return qr_fail(rc, err);
    const int n_srf = (int)v.hdr->n_srf, n_lgt = (int)v.hdr->n_lgt;
    for (int i = 0; i < n; i++)
    {
        const qr_node &nd = next[i];
        const int refs[3] = { nd.srf, nd.tag == QR_NODE_ARRAY ? nd.inb : -1, nd.tag == QR_NODE_ARRAY ? nd.bvb : -1 };
        for (int r : refs) if (r < -1 || r >= n_srf) return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": surface record out of range");
        if (nd.tag == QR_NODE_LIGHT && (nd.lgt < -1 || nd.lgt >= n_lgt)) return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": light record out of range");
        if (st[(size_t)i].trnode >= 0 && st[(size_t)i].trnode != i && (is_surface(nd.tag) || nd.tag == QR_NODE_ARRAY) && nd.srf >= 0
            && next[st[(size_t)i].trnode].srf < 0 && base == nullptr)
            return qr_fail(QR_ERR_UNSUP, "node " + std::to_string(i) + ": its transform node has no record in the snapshot");
    }
    if (camera >= n || (camera >= 0 && next[camera].tag != QR_NODE_CAMERA)) return qr_fail(QR_ERR_ARG, "camera is not a camera node");

    if (base != nullptr)
    {
        /* the scope of a frame-to-frame update (qr_hierarchy.h): same transform nodes, same axis mappings and scalers,
         * no moving bounding volume */
        st0.resize((size_t)n);
        rc = run_update(base, n, opts, st0.data(), err);
        if (rc != QR_OK) return qr_fail(rc, err);
        std::vector<char> moved((size_t)n, 0);
        for (int i = 0; i < n; i++)
        {
            const qr_node_state &a = st0[(size_t)i], &b = st[(size_t)i];
            if (base[i].parent != next[i].parent || base[i].tag != next[i].tag || base[i].srf != next[i].srf || base[i].lgt != next[i].lgt)
                return qr_fail(QR_ERR_ARG, "node " + std::to_string(i) + ": the two node tables describe different trees");
            const bool has_record = (is_surface(next[i].tag) || next[i].tag == QR_NODE_ARRAY) && next[i].srf >= 0;
            if (has_record && (a.trnode != b.trnode || a.obj_has_trm != b.obj_has_trm))
            {
                /* The set of transform nodes changes.  The global list holds one element per ARRAY that is the transform node of
                 * surfaces, in front of its members, which stand together (rt_SceneThread::insert, engine.cpp:1148-1214; bounding-
                 * volume arrays are not in a camera list when the screen is tiled, 1711-1725): when surfaces change their array
                 * -- the rotating light arrays of the demo scenes (scn_demo01.h:513-561), an array under a turning array that
                 * starts to turn itself, any of them stopping -- the list is regrouped below and records of new transform nodes
                 * are created.  A surface that becomes (or stops being) its OWN transform node changes no element: its record
                 * carries the matrix itself.  Custom clipping is where it ends: clippers' lists carry transform-node markers of
                 * their own (engine.cpp:1845-1947) */
                if (is_surface(next[i].tag))
                {
                    const int g_old = a.trnode >= 0 && a.trnode != i ? a.trnode : -1, g_new = b.trnode >= 0 && b.trnode != i ? b.trnode : -1;
                    if (g_old != g_new) regroup = true;
                    if (g_old != g_new || (a.trnode == i) != (b.trnode == i)) self_of.push_back(i);
                    if (g_new >= 0 && next[g_new].tag != QR_NODE_ARRAY)
                        return qr_fail(QR_ERR_UNSUP, "node " + std::to_string(i) + ": its transform node is not an array");
                }
            }
            if (has_record && !(flags & QR_HIER_BOUNDS) &&
                (!same_bits(a.map, b.map, sizeof(a.map)) || !same_bits(a.sgn, b.sgn, sizeof(a.sgn)) || !same_bits(a.scl, b.scl, sizeof(a.scl))))
                return qr_fail(QR_ERR_UNSUP, "node " + std::to_string(i) + ": axis mapping or scalers change (clip boxes would; QR_HIER_BOUNDS recomputes them)");
            moved[(size_t)i] = !same_bits(a.mtx, b.mtx, sizeof(a.mtx));
            if (has_record && next[i].tag == QR_TAG_PLANE)
            {
                /* a plane's axis scalers enter the texture scale and offset of its two materials (rt_Plane::update_fields,
                 * object.cpp:2893-2938): asc = scl[mp_i], scl[mp_j].  The material's own scale and position are not in the
                 * snapshot; they can be read back from its record exactly where the base scalers are 1 (x * (1/1), x * 1) */
                const float a0[2] = { a.scl[a.map[0]], a.scl[a.map[1]] }, a1[2] = { b.scl[b.map[0]], b.scl[b.map[1]] };
                bool textured = false;                      /* a one-texel texture is read at (0, 0) whatever the scale */
                for (int side = 0; side < 2; side++)
                {
                    const int32_t mi = v.srf[next[i].srf].mat[side];
                    if (mi != QR_NULL && (v.mat[mi].xmask != 0 || v.mat[mi].ymask != 0)) textured = true;
                }
                if (textured && !same_bits(a0, a1, sizeof(a0)))
                {
                    if (!next[i].has_tex && (a0[0] != 1.0f || a0[1] != 1.0f))
                        return qr_fail(QR_ERR_UNSUP, "node " + std::to_string(i) + ": a scaled plane's axis scalers change (its materials' texture scale is not recoverable from the snapshot: give qr_node.tex)");
                    plane_tex.push_back(i);
                }
            }
        }
        for (int i = n - 1; i >= 0; i--)
        {
            if (moved[(size_t)i] && next[i].parent >= 0) moved[(size_t)next[i].parent] = 1;  /* children come later: one sweep */
            if (next[i].tag == QR_NODE_ARRAY && moved[(size_t)i] && (next[i].inb >= 0 || next[i].bvb >= 0) && !(flags & QR_HIER_BOUNDS))
                return qr_fail(QR_ERR_UNSUP, "node " + std::to_string(i) + ": an array with a bounding volume moves (QR_HIER_BOUNDS recomputes the volumes)");
        }
    }

    /* ---- surfaces change their array: records of new transform nodes, and the global list regrouped ---- */
    std::vector<qr_node> nodes_w;                           /* `next` with the records of new transform nodes entered */
    std::vector<qr_surface> S_w;
    std::vector<qr_elem> E_w;
    int32_t clist_w = v.frame->clist;
    /* the engine removes surfaces that are fully hidden behind others from its camera list (RT_OPTS_REMOVE, the 4|x results of
     * bbox_sort, rtgeom.cpp:1551-1560): a list that lacks surfaces of the table is only good for the frame it was made for */
    bool incomplete = false;
    if (base != nullptr && (flags & QR_HIER_RESET_TILES))
    {
        size_t in_list = 0, in_table = 0;
        for (int32_t e = clist_w; e != QR_NULL; e = v.elm[e].next) if (v.elm[e].data == QR_NULL) in_list++;
        for (int i = 0; i < n; i++) if (is_surface(next[i].tag) && next[i].srf >= 0) in_table++;
        incomplete = in_list < in_table;
    }
    if (base != nullptr && (flags & QR_HIER_REGROUP)) regroup = true;
    const bool restructure = regroup || incomplete;
    if (restructure)
    {
        if (!(flags & QR_HIER_RESET_TILES))
            return qr_fail(QR_ERR_ARG, "the set of transform nodes changes: the tile lists go stale, pass QR_HIER_RESET_TILES (and rebuild the lists)");
        if (regroup && !(flags & QR_HIER_BOUNDS))
            return qr_fail(QR_ERR_ARG, "the set of transform nodes changes: the members' boxes move to another space, pass QR_HIER_BOUNDS (node tables with the bounds inputs)");
        nodes_w.assign(next, next + n);
        S_w.assign(v.srf, v.srf + v.hdr->n_srf);
        E_w.assign(v.elm, v.elm + v.hdr->n_elm);
        std::vector<int> node_of_rec((size_t)n_srf, -1), head_of_rec((size_t)n_srf, -1);
        for (int i = 0; i < n; i++)
        {
            if (is_surface(next[i].tag) && next[i].srf >= 0) node_of_rec[(size_t)next[i].srf] = i;
            if (next[i].tag == QR_NODE_ARRAY && next[i].srf >= 0) head_of_rec[(size_t)next[i].srf] = i;
        }
        /* the leaves of the list in its order; array elements must be what the base state says they are */
        std::vector<int> leaf;                              /* node of every leaf, in list order */
        for (int32_t e = clist_w; e != QR_NULL; e = v.elm[e].next)
        {
            const qr_elem &el = v.elm[e];
            if (el.data != QR_NULL)
            {
                const int arr = head_of_rec[(size_t)el.simd];
                if (arr < 0 || el.kind != 0 || st0[(size_t)arr].trnode != arr)
                    return qr_fail(QR_ERR_UNSUP, "the global list holds an array element that is not a transform node of the base table: cannot regroup it");
                continue;
            }
            const int nd = node_of_rec[(size_t)el.simd];
            if (nd < 0) return qr_fail(QR_ERR_UNSUP, "the global list holds a surface no node of the table owns: cannot regroup it");
            leaf.push_back(nd);
        }
        if (incomplete)
        {
            /* surfaces the engine had removed come back behind the others */
            std::vector<char> listed((size_t)n, 0);
            for (int nd : leaf) listed[(size_t)nd] = 1;
            for (int i = 0; i < n; i++)
                if (is_surface(next[i].tag) && next[i].srf >= 0 && !listed[(size_t)i]) leaf.push_back(i);
        }
        auto group_of = [&](int nd) { const int t = st[(size_t)nd].trnode; return t >= 0 && t != nd ? t : -1; };
        /* records of arrays that become transform nodes (rt_Array's s_srf as rt_Node::update_fields fills it, object.cpp:813-843:
         * tag -1, no solver, its own transform node); an array that was one before still has its record */
        std::vector<char> heads_surfaces((size_t)n, 0);
        for (int nd = 0; nd < n; nd++)
            if (is_surface(next[nd].tag) && next[nd].srf >= 0 && group_of(nd) >= 0) heads_surfaces[(size_t)group_of(nd)] = 1;
        for (int g = 0; g < n; g++)                         /* in node order: the k-th new record is n_srf + k (qr_hierarchy.h) */
        {
            if (!heads_surfaces[(size_t)g] || nodes_w[(size_t)g].srf >= 0) continue;
            qr_surface rec; memset(&rec, 0, sizeof(rec));
            rec.c_def = 0xFFFFFFFFu; rec.smask = 0x80000000u; rec.d_eps = 1e-11f; rec.t_eps = 1e-7f;       /* the constants of every record: object.h:41-42 */
            rec.srf_t[3] = QR_NODE_ARRAY; rec.clip = QR_NULL; rec.trnode = (int32_t)S_w.size();
            rec.mat[0] = rec.mat[1] = QR_NULL; for (int k = 0; k < 4; k++) rec.lst[k] = QR_NULL;
            nodes_w[(size_t)g].srf = (int32_t)S_w.size();
            S_w.push_back(rec);
        }
        /* the new list, in fresh elements (the old chain may share cells with other lists): every group where its first member
         * stood -- [array element, members in list order], `data` of the array element = the last member -- other leaves alone */
        std::vector<char> placed(leaf.size(), 0);
        int32_t tail = QR_NULL; clist_w = QR_NULL;
        auto append = [&](int32_t simd) -> int32_t {
            qr_elem ne; ne.simd = simd; ne.data = QR_NULL; ne.next = QR_NULL; ne.kind = 0;
            const int32_t ix = (int32_t)E_w.size();
            E_w.push_back(ne);
            if (tail == QR_NULL) clist_w = ix; else E_w[(size_t)tail].next = ix;
            tail = ix;
            return ix;
        };
        for (size_t k = 0; k < leaf.size(); k++)
        {
            if (placed[k]) continue;
            const int g = group_of(leaf[k]);
            if (g < 0) { append(next[leaf[k]].srf); placed[k] = 1; continue; }
            const int32_t head = append(nodes_w[(size_t)g].srf);
            for (size_t m = k; m < leaf.size(); m++)
                if (!placed[m] && group_of(leaf[m]) == g) { E_w[(size_t)head].data = append(next[leaf[m]].srf); placed[m] = 1; }
        }
        /* clipper lists (engine.cpp:1845-1947): accum markers (no record; data -1 / +1), clippers (data = the side that clips) and
         * transform-node markers (kind 2, data = the last clipper under them) in front of the clippers that share an array -- the
         * same regrouping, stretch by stretch between the accum markers.  The order of clippers inside a stretch is free (every
         * one of them must pass / any of an accum segment); lists none of whose clippers changes its array stay as they are */
        auto group_before = [&](int nd) { const int t = st0[(size_t)nd].trnode; return t >= 0 && t != nd ? t : -1; };
        std::vector<std::pair<int32_t, int32_t>> redone;    /* (old head, new head): lists are shared between records */
        for (size_t q = 0; q < (size_t)n_srf; q++)
        {
            const int32_t h = v.srf[q].clip;
            if (h == QR_NULL) continue;
            size_t k = 0;
            while (k < redone.size() && redone[k].first != h) k++;
            if (k < redone.size()) { S_w[q].clip = redone[k].second; continue; }
            bool changes = false;
            for (int32_t e = h; e != QR_NULL; e = v.elm[e].next)
            {
                const qr_elem &el = v.elm[e];
                if (el.simd == QR_NULL || el.kind == 2 || v.srf[el.simd].srf_t[3] < 0) continue;
                const int nd = node_of_rec[(size_t)el.simd];
                if (nd < 0) return qr_fail(QR_ERR_UNSUP, "a clipper list holds a surface no node of the table owns: cannot regroup it");
                if (group_before(nd) != group_of(nd) || (flags & QR_HIER_REGROUP)) changes = true;
            }
            int32_t nh = h;
            if (changes)
            {
                int32_t ltail = QR_NULL; nh = QR_NULL;
                auto put = [&](const qr_elem &src) -> int32_t {
                    qr_elem ne = src; ne.next = QR_NULL;
                    const int32_t ix = (int32_t)E_w.size();
                    E_w.push_back(ne);
                    if (ltail == QR_NULL) nh = ix; else E_w[(size_t)ltail].next = ix;
                    ltail = ix;
                    return ix;
                };
                std::vector<int32_t> run;                   /* clippers of the stretch being collected */
                auto flush = [&]() {
                    std::vector<char> done(run.size(), 0);
                    for (size_t a = 0; a < run.size(); a++)
                    {
                        if (done[a]) continue;
                        const int g = group_of(node_of_rec[(size_t)v.elm[run[a]].simd]);
                        if (g < 0) { put(v.elm[run[a]]); done[a] = 1; continue; }
                        qr_elem mk; mk.simd = nodes_w[(size_t)g].srf; mk.data = QR_NULL; mk.next = QR_NULL; mk.kind = 2;
                        const int32_t head = put(mk);
                        for (size_t b = a; b < run.size(); b++)
                            if (!done[b] && group_of(node_of_rec[(size_t)v.elm[run[b]].simd]) == g) { E_w[(size_t)head].data = put(v.elm[run[b]]); done[b] = 1; }
                    }
                    run.clear();
                };
                for (int32_t e = h; e != QR_NULL; e = v.elm[e].next)
                {
                    const qr_elem &el = v.elm[e];
                    if (el.simd != QR_NULL && (el.kind == 2 || v.srf[el.simd].srf_t[3] < 0)) continue;
                    if (el.simd == QR_NULL) { flush(); put(el); }
                    else run.push_back(e);
                }
                flush();
            }
            redone.push_back(std::make_pair(h, nh));
            S_w[q].clip = nh;
        }
        next = nodes_w.data();
    }

    /* the snapshot that is patched: a copy of the input, or -- when records and elements were added -- a re-serialised one */
    std::vector<uint8_t> out;
    qr_header hdr_w = *v.hdr;
    if (!restructure) out.assign((const uint8_t *)blob, (const uint8_t *)blob + v.hdr->total_bytes);
    else
    {
        auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
        size_t off = a16(sizeof(qr_header));
        hdr_w.header_bytes = sizeof(qr_header);
        hdr_w.n_srf = (uint32_t)S_w.size(); hdr_w.n_elm = (uint32_t)E_w.size();
        hdr_w.off_frame = (uint32_t)off;  off = a16(off + sizeof(qr_frame));
        hdr_w.off_srf = (uint32_t)off;    off = a16(off + S_w.size() * sizeof(qr_surface));
        hdr_w.off_mat = (uint32_t)off;    off = a16(off + (size_t)hdr_w.n_mat * sizeof(qr_material));
        hdr_w.off_lgt = (uint32_t)off;    off = a16(off + (size_t)hdr_w.n_lgt * sizeof(qr_light));
        hdr_w.off_elm = (uint32_t)off;    off = a16(off + E_w.size() * sizeof(qr_elem));
        hdr_w.off_tiles = (uint32_t)off;  off = a16(off + (size_t)hdr_w.n_tiles * 4);
        hdr_w.off_texels = (uint32_t)off; off = a16(off + (size_t)hdr_w.n_texels * 4);
        hdr_w.total_bytes = (uint32_t)off;
        out.assign(off, 0);
        memcpy(out.data(), &hdr_w, sizeof(hdr_w));
        memcpy(out.data() + hdr_w.off_frame, v.frame, sizeof(qr_frame));
        memcpy(out.data() + hdr_w.off_srf, S_w.data(), S_w.size() * sizeof(qr_surface));
        if (hdr_w.n_mat) memcpy(out.data() + hdr_w.off_mat, v.mat, (size_t)hdr_w.n_mat * sizeof(qr_material));
        if (hdr_w.n_lgt) memcpy(out.data() + hdr_w.off_lgt, v.lgt, (size_t)hdr_w.n_lgt * sizeof(qr_light));
        memcpy(out.data() + hdr_w.off_elm, E_w.data(), E_w.size() * sizeof(qr_elem));
        if (hdr_w.n_tiles) memcpy(out.data() + hdr_w.off_tiles, v.tiles, (size_t)hdr_w.n_tiles * 4);
        if (hdr_w.n_texels) memcpy(out.data() + hdr_w.off_texels, v.texels, (size_t)hdr_w.n_texels * 4);
        ((qr_frame *)(out.data() + hdr_w.off_frame))->clist = clist_w;
    }
    qr_surface *S = (qr_surface *)(out.data() + hdr_w.off_srf);
    qr_light *L = (qr_light *)(out.data() + hdr_w.off_lgt);
    qr_frame *F = (qr_frame *)(out.data() + hdr_w.off_frame);
    for (int i = 0; i < n; i++)
    {
        const qr_node &nd = next[i];
        const qr_node_state &s = st[(size_t)i];
        const M4 &m = *(const M4 *)s.mtx;
        if ((is_surface(nd.tag) || nd.tag == QR_NODE_ARRAY) && nd.srf >= 0)
        {
            qr_surface &r = S[nd.srf];
            /* rt_Node::update_fields 813-843: transform nodes carry the inverse matrix; everybody its position */
            if (s.trnode == i)
            {
                M4 inv;
                inverse3(inv, m);
                for (int c = 0; c < 3; c++) { r.tci[c] = inv[c][0]; r.tcj[c] = inv[c][1]; r.tck[c] = inv[c][2]; }
            }
            r.pos[0] = m[3][0]; r.pos[1] = m[3][1]; r.pos[2] = m[3][2];
            put_mapping(r, s, next);
            if (is_surface(nd.tag)) put_quadric(r, nd, s);
        }
        /* the records of an array's bounding volumes (nd.inb, nd.bvb) are left alone: rt_Array::update_bounds
         * (object.cpp:1830-2316) owns them and overrides what update_fields wrote (2268-2280) */
        if (nd.tag == QR_NODE_LIGHT && nd.lgt >= 0)
        {
            /* rt_Light::update_fields 649-667 */
            L[nd.lgt].pos[0] = m[3][0]; L[nd.lgt].pos[1] = m[3][1]; L[nd.lgt].pos[2] = m[3][2];
        }
    }
    for (int i : self_of)
        if (st[(size_t)i].trnode != i && st0[(size_t)i].trnode == i)
        {
            /* not its own transform node any more: the engine's record of such a surface holds no matrix (the field is never read) */
            qr_surface &r = S[next[i].srf];
            for (int c = 0; c < 3; c++) r.tci[c] = r.tcj[c] = r.tck[c] = 0.0f;
        }
    if (!plane_tex.empty())
    {
        qr_material *M = (qr_material *)(out.data() + hdr_w.off_mat);
        for (int i : plane_tex)
        {
            const qr_node_state &b = st[(size_t)i];
            const float asc[2] = { b.scl[b.map[0]], b.scl[b.map[1]] };
            const float isc[2] = { 1.0f / asc[0], 1.0f / asc[1] };
            const int32_t *mats = v.srf[next[i].srf].mat;
            for (int side = 0; side < 2; side++)
            {
                const int32_t mi = mats[side];
                if (mi == QR_NULL || (side == 1 && mi == mats[0])) continue;
                const qr_material &m0 = v.mat[mi];
                if ((uint32_t)m0.t_map[0] > 1u || (uint32_t)m0.t_map[1] > 1u) return qr_fail(QR_ERR_ARG, "material: texture axis out of range");
                /* the material's own scale and position: given in the node table, or read back from the record (base scalers 1) */
                const float *tx = next[i].tex + 4 * side;
                const float sx = next[i].has_tex ? tx[0] : m0.xscal, sy = next[i].has_tex ? tx[1] : m0.yscal;
                const float px = next[i].has_tex ? tx[2 + m0.t_map[0]] : m0.xoffs, py = next[i].has_tex ? tx[2 + m0.t_map[1]] : m0.yoffs;
                M[mi].xscal = sx * isc[m0.t_map[0]];
                M[mi].yscal = sy * isc[m0.t_map[1]];
                M[mi].xoffs = px * asc[m0.t_map[0]];
                M[mi].yoffs = py * asc[m0.t_map[1]];
            }
        }
    }
    if (flags & QR_HIER_BOUNDS)
    {
        /* clip boxes of surfaces and the records of arrays' bounding volumes for the new transforms (qr_hbounds.cpp) */
        std::vector<qr_node_bounds> nb((size_t)n);
        rc = restructure ? qr_hierarchy_bounds(out.data(), out.size(), next, n, opts, nb.data()) : qr_hierarchy_bounds(blob, size, next, n, opts, nb.data());
        if (rc != QR_OK) return rc;
        for (int i = 0; i < n; i++)
        {
            const qr_node &nd = next[i];
            const qr_node_state &s = st[(size_t)i];
            const qr_node_bounds &b = nb[(size_t)i];
            if (is_surface(nd.tag) && nd.srf >= 0)
            {
                /* rt_Surface::update_bounds 2824-2844: which sides clip, and the box relative to the position */
                qr_surface &r = S[nd.srf];
                uint32_t mt = 0;
                for (int x = 0; x < 3; x++)
                {
                    const float pp = s.trnode == i ? 0.0f : s.mtx[12 + x];
                    if (b.cmin[x] != -FLT_MAX) mt |= 1u << x;
                    if (b.cmax[x] != +FLT_MAX) mt |= 1u << (3 + x);
                    r.min[x] = b.bmin[x] - pp;
                    r.max[x] = b.bmax[x] - pp;
                }
                r.minmax_t = mt;
            }
            if (nd.tag != QR_NODE_ARRAY) continue;
            /* rt_Array::update_bounds 2181-2310: the records of the inner and outer bounding volume */
            auto ellipsoid = [&](qr_surface &r, const float *lo, const float *hi) {
                for (int x = 0; x < 3; x++) r.pos[x] = (lo[x] + hi[x]) * 0.5f;
                r.sci[3] = 0.75f;                                       /* the unit cube's radius squared */
                for (int x = 0; x < 3; x++) { const float d = hi[x] - lo[x]; r.sci[x] = 1.0f / (d * d); }
            };
            if (nd.inb >= 0 && b.inb_form == 1)
            {
                ellipsoid(S[nd.inb], b.inmin, b.inmax);
                put_mapping(S[nd.inb], s, next);                        /* rt_Array::update_fields 1790-1802: the array's own mapping */
            }
            if (nd.inb >= 0 && b.inb_form == 2)
            {
                qr_surface &r = S[nd.inb];
                r.axes = 0u | 1u << 2 | 2u << 4;                         /* identity axis map, no signs; no transform node of its own */
                r.has_trm = 0; r.shift = 0; r.trnode = QR_NULL;
                for (int x = 0; x < 3; x++) { r.pos[x] = b.inmid[x]; r.sci[x] = 1.0f; }
                r.sci[3] = b.inrad * b.inrad;
            }
            if (nd.bvb >= 0 && b.bvb_form == 1) ellipsoid(S[nd.bvb], b.bmin, b.bmax);
        }
    }
    if (camera >= 0)
    {
        /* rt_Scene::render, engine.cpp:3029-3043, 3256-3260: rays start at the camera position and aim at the centre
         * of the top-left pixel of a screen `pov` in front of it; hor / ver step one pixel */
        const M4 &m = *(const M4 *)st[(size_t)camera].mtx;
        const float factor = 1.0f / (float)F->frm_w, aspect = (float)F->frm_h * factor;
        const float h = -0.5f * 1.0f, w = -0.5f * aspect, pov = next[camera].pov;
        float dir[3], hor[3], ver[3];
        for (int c = 0; c < 3; c++)
        {
            dir[c] = m[2][c] * pov;
            dir[c] += m[0][c] * h;
            dir[c] += m[1][c] * w;
            hor[c] = m[0][c] * factor;
            ver[c] = m[1][c] * factor;
            dir[c] += hor[c] * 0.5f;
            dir[c] += ver[c] * 0.5f;
            F->dir[c] = dir[c]; F->hor[c] = hor[c]; F->ver[c] = ver[c]; F->org[c] = m[3][c];
        }
    }
    if (flags & QR_HIER_RESET_TILES)
    {
        int32_t *T = (int32_t *)(out.data() + hdr_w.off_tiles);
        for (uint32_t k = 0; k < hdr_w.n_tiles; k++) T[k] = F->clist;
    }
    void *p = malloc(out.size());
    if (p == nullptr) return qr_fail(QR_ERR_NOMEM, "out of memory");
    memcpy(p, out.data(), out.size());
    *out_blob = p;
    *out_size = out.size();
    return QR_OK;
}
