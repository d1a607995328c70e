/*
 * qr_compile.cpp - host side: snapshot (include/qr_scene.h) -> the compiled device image of qr_program.h.
 *
 *   qr_snapshot_validate   every index, list and tag the compiler follows (a malformed snapshot is rejected
 *                          here and can never become an out-of-bounds device access)
 *   qr_program_build       per-surface records, list programs, clipper programs, light lists, wave schedule
 *   qr_program_verify      walks the finished image once more and checks every offset the kernel will follow
 *
 * What the list compiler resolves statically is the reference's per-ray walk state that depends only on the
 * list position: `ctx_LOCAL(OBJ)` (tracer.cpp:1385-1417: are we inside a trnode whose transform is cached?),
 * the end of bounding-volume arrays (AR_skp 3955-4054), and, for clipper lists, the same caching of the clipper
 * trnode (1931-2151).  Pure host code: no HIP here, so the CPU test-suite exercises it on every fixture.
 */
#include "qr_internal.h"
#include "qr_sides.h"
#include "qr_program.h"
#include "qr_bounds.hpp"

#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <thread>

namespace {

inline bool is_real(const qr_surface &s) { return s.srf_t[3] >= 0 && s.srf_t[3] < QR_TAG_SURFACE_MAX; }

struct Fail { int rc; std::string msg; };

} // namespace

/* ------------------------------------------------------------------------------------------------------- */
/* validation                                                                                              */
/* ------------------------------------------------------------------------------------------------------- */

int qr_snapshot_validate(const qr_scene_view &v, std::string &err)
{
    const qr_frame &fr = *v.frame;
    auto bad = [&](int rc, const char *m) { err = m; return rc; };
    if (fr.fsaa < 0 || fr.fsaa > 2) return bad(QR_ERR_UNSUP, "unsupported fsaa");
    if (fr.frm_w <= 0 || fr.frm_h <= 0 || fr.tile_w <= 0 || fr.tile_h <= 0) return bad(QR_ERR_ARG, "bad frame parameters");
    if (fr.depth < 0) return bad(QR_ERR_ARG, "negative recursion depth");
    if (fr.tls_row <= 0 || fr.tls_col <= 0) return bad(QR_ERR_ARG, "tile grid must be positive");
    if ((int64_t)fr.tls_row * fr.tile_w < fr.frm_w || (int64_t)fr.tls_col * fr.tile_h < fr.frm_h)
        return bad(QR_ERR_ARG, "tile grid does not cover the frame");
    if (fr.thnum < 0 || fr.index < 0 || (fr.thnum > 0 && fr.index >= fr.thnum) || (fr.thnum == 0 && fr.index != 0))
        return bad(QR_ERR_ARG, "bad index/thnum");

    const int n_srf = (int)v.hdr->n_srf, n_mat = (int)v.hdr->n_mat, n_lgt = (int)v.hdr->n_lgt;
    const int n_elm = (int)v.hdr->n_elm, n_tex = (int)v.hdr->n_texels;
    auto ok_elm = [&](int i) { return i == QR_NULL || (i >= 0 && i < n_elm); };
    auto ok_srf = [&](int i) { return i >= 0 && i < n_srf; };
    for (int i = 0; i < n_elm; i++)
        if (!ok_elm(v.elm[i].next)) return bad(QR_ERR_ARG, "element next out of range");
    for (uint32_t i = 0; i < v.hdr->n_tiles; i++)
        if (!ok_elm(v.tiles[i])) return bad(QR_ERR_ARG, "tile head out of range");
    if (!ok_elm(fr.clist)) return bad(QR_ERR_ARG, "clist out of range");
    for (int i = 0; i < n_mat; i++)
    {
        const qr_material &m = v.mat[i];
        const uint64_t n = (uint64_t)(m.xmask + 1) * (m.ymask + 1);
        if (m.tex < 0 || (uint64_t)m.tex + n > (uint64_t)n_tex) return bad(QR_ERR_ARG, "texture out of range");
        if (m.t_map[0] < 0 || m.t_map[0] > 1 || m.t_map[1] < 0 || m.t_map[1] > 1) return bad(QR_ERR_ARG, "bad t_map");
        if ((m.xmask & (m.xmask + 1)) != 0 || (m.ymask & (m.ymask + 1)) != 0) return bad(QR_ERR_ARG, "texture size not a power of two");
        if (((uint64_t)m.ymask << (m.yshft & 31)) + m.xmask >= n) return bad(QR_ERR_ARG, "texture addressing exceeds texture");
    }
    /* every list is walked once per kind with a step bound (cycle check) */
    std::vector<uint8_t> checked((size_t)n_elm + 1, 0);
    auto check_list = [&](int head, int kind) -> const char * {
        /* kind 0 surfaces, 1 clippers, 2 lights */
        if (head == QR_NULL) return nullptr;
        if (checked[head] & (1u << kind)) return nullptr;
        checked[head] |= (uint8_t)(1u << kind);
        int cnt = 0;
        for (int e = head; e != QR_NULL; e = v.elm[e].next)
        {
            if (++cnt > n_elm) return "cyclic list";
            const qr_elem &el = v.elm[e];
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
            else if (el.simd != QR_NULL)
            {
                if (!ok_srf(el.simd)) return "clipper index out of range";
                if (v.srf[el.simd].srf_t[3] < 0 && !ok_elm(el.data)) return "clip trnode last out of range";
            }
        }
        return nullptr;
    };
    for (uint32_t i = 0; i < v.hdr->n_tiles; i++)
        if (const char *m = check_list(v.tiles[i], 0)) return bad(QR_ERR_ARG, m);
    if (const char *m = check_list(fr.clist, 0)) return bad(QR_ERR_ARG, m);
    for (int i = 0; i < n_srf; i++)
    {
        const qr_surface &s = v.srf[i];
        if (s.trnode != QR_NULL && !ok_srf(s.trnode)) return bad(QR_ERR_ARG, "trnode out of range");
        const bool real = is_real(s);
        if (s.has_trm != 0 && s.trnode == QR_NULL && real) return bad(QR_ERR_ARG, "transformed surface without trnode");
        for (int k = 0; k < 3; k++)
            if (((s.axes >> (2 * k)) & 3) > 2) return bad(QR_ERR_ARG, "bad axis map");
        if ((s.conic & ~3) || (s.has_trm & ~3) || (s.srf_t[0] & ~3) || (s.srf_t[1] & ~3) || (s.srf_t[2] & ~3))
            return bad(QR_ERR_ARG, "surface tag fields out of range");
        if (!real) continue;
        if (s.smask != 0x80000000u) return bad(QR_ERR_ARG, "surface smask is not the fp32 sign bit");
        if ((s.shift != 0) != (s.has_trm != 0))
            return bad(QR_ERR_UNSUP, "surface with trnode shift but no transform flags (or the reverse)");
        for (int k = 0; k < 2; k++)
            if (s.mat[k] < 0 || s.mat[k] >= n_mat) return bad(QR_ERR_ARG, "material index out of range");
        if (!ok_elm(s.clip) || !ok_elm(s.lst[0]) || !ok_elm(s.lst[1]) || !ok_elm(s.lst[2]) || !ok_elm(s.lst[3]))
            return bad(QR_ERR_ARG, "surface list head out of range");
        if (const char *m = check_list(s.clip, 1)) return bad(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[0], 2)) return bad(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[2], 2)) return bad(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[1], 0)) return bad(QR_ERR_ARG, m);
        if (const char *m = check_list(s.lst[3], 0)) return bad(QR_ERR_ARG, m);
        for (int side = 0; side < 2; side++)
            for (int e = s.lst[side * 2]; e != QR_NULL; e = v.elm[e].next)
                if (const char *m = check_list(v.elm[e].data, 0)) return bad(QR_ERR_ARG, m);
    }
    return QR_OK;
}

void qr_bound_spheres(const qr_scene_view &v, std::vector<BSphere> &out)
{
    const int n = (int)v.hdr->n_srf;
    out.resize((size_t)n + 1);
    for (int i = 0; i < n; i++)
    {
        out[i] = bound_sphere(v, i);
    }
    out[n].c[0] = out[n].c[1] = out[n].c[2] = 0.0f; out[n].r = __builtin_inff();
}

/* ------------------------------------------------------------------------------------------------------- */
/* the compiler                                                                                            */
/* ------------------------------------------------------------------------------------------------------- */

namespace {

#ifndef QR_FLAT_LIST_MAX
#define QR_FLAT_LIST_MAX 128
#endif

struct ListFilter
{
    const qr_scene_view &v;
    const std::vector<BSphere> &bs;
    std::vector<qr_elem> &E;                /* grows */
    struct Node { int e; int si; int last; bool head; BSphere bound; bool bounded; };
    std::vector<Node> ch;
    int n_real = 0;                         /* surfaces in the chain */

    /* the chain at `clist` of E (which also receives the filtered chains: elements are read by index only) */
    ListFilter(const qr_scene_view &v_, const std::vector<BSphere> &bs_, std::vector<qr_elem> &E_, int clist)
        : v(v_), bs(bs_), E(E_)
    {
        std::vector<int> pos_of(E.size(), -1);
        for (int e = clist; e != QR_NULL; e = E[e].next) { pos_of[e] = (int)ch.size(); ch.push_back(Node{e, E[e].simd, -1, false, {}, false}); }
        const int n = (int)ch.size();
        n_real = 0;
        for (int i = 0; i < n; i++) if (is_real(v.srf[ch[i].si])) n_real++;
        for (int i = 0; i < n; i++)
        {
            const qr_elem el = E[ch[i].e];
            const qr_surface &s = v.srf[el.simd];
            const bool head = (el.kind & 3) == 1 || s.srf_t[3] < 0;
            ch[i].head = head && el.data != QR_NULL && pos_of[el.data] >= i;
            if (ch[i].head) ch[i].last = pos_of[el.data];
            else { ch[i].bound = bs[el.simd]; ch[i].bounded = is_real(s) && bs[el.simd].r < 1e18f; }
        }
        /* union spheres of arrays, innermost first (a later head nests inside an earlier one) */
        for (int i = n - 1; i >= 0; i--)
        {
            if (!ch[i].head) continue;
            bool ok = true; int first = -1;
            for (int k = i + 1; k <= ch[i].last; k++)
                if (!ch[k].head) { if (!is_real(v.srf[ch[k].si])) continue; if (!ch[k].bounded) ok = false; else if (first < 0) first = k; }
            ch[i].bounded = false;
            if (!ok || first < 0) continue;
            const BSphere &c0 = ch[first].bound;
            double R = 0.0;
            for (int k = i + 1; k <= ch[i].last; k++)
                if (!ch[k].head && ch[k].bounded)
                {
                    const BSphere &m = ch[k].bound;
                    const double dx = (double)m.c[0] - c0.c[0], dy = (double)m.c[1] - c0.c[1], dz = (double)m.c[2] - c0.c[2];
                    const double d = __builtin_sqrt(dx * dx + dy * dy + dz * dz) + (double)m.r;
                    if (d > R) R = d;
                }
            ch[i].bound = c0; ch[i].bound.r = (float)(R * 1.0001 + 1e-4);
            ch[i].bounded = true;
        }
    }

    /*
     * A filtered chain before it is part of E: its elements with `next` (and the `data` of kept array heads, flagged in
     * `local`) as indices INTO THE RUN.  filter_run only reads the filter and E, so many threads may filter one chain at a time;
     * append() then makes a run part of E, one after the other.
     */
    struct Run { std::vector<qr_elem> el; std::vector<uint8_t> local; };
    enum { SOURCE_LIST = -2 };              /* filter_run's result when nothing was dropped: the source chain itself */

    /* keep(sphere, bounded) -> may a surface with this bound matter; returns the new list's head */
    template <typename Pred>
    int filter(Pred keep)
    {
        Run run;
        return append(run, filter_run(keep, run));
    }
    template <typename Pred>
    int filter_run(Pred keep, Run &run) const
    {
        return filter_nodes_run([&](int i) {
            const Node &nd = ch[i];
            if (nd.head) return !nd.bounded || keep(nd.bound, true);
            return keep(nd.bound, nd.bounded);
        }, run);
    }

    /* keep(i) by position in the chain: for an array element "may any member matter" (false prunes the array without
     * visiting its members), for a surface "does it belong on the list".  A list that keeps everything IS the source list:
     * its head is returned and nothing is allocated (per-surface copies of one global list are what makes the engine's
     * lists grow with the square of the scene). */
    template <typename Pred>
    int filter_nodes(Pred keep)
    {
        Run run;
        return append(run, filter_nodes_run(keep, run));
    }

    /* the run becomes the tail of E; head as filter_nodes_run returned it -> the chain's head in E */
    int append(const Run &run, int head)
    {
        if (head == SOURCE_LIST) return ch[0].e;
        const int base = (int)E.size();
        for (size_t i = 0; i < run.el.size(); i++)
        {
            qr_elem c = run.el[i];
            if (c.next != QR_NULL) c.next += base;
            if (run.local[i] && c.data != QR_NULL) c.data += base;
            E.push_back(c);
        }
        return head == QR_NULL ? QR_NULL : head + base;
    }

    template <typename Pred>
    int filter_nodes_run(Pred keep, Run &run) const
    {
        run.el.clear(); run.local.clear();
        std::vector<qr_elem> &R = run.el;
        bool dropped = false;
        struct Open { int idx; int out; int last; };
        std::vector<Open> open;
        int head = QR_NULL, tail = QR_NULL;
        auto emit = [&](int src_e, int data) {
            qr_elem c = E[src_e];
            c.data = data; c.next = QR_NULL;
            R.push_back(c); run.local.push_back(0);
            const int ix = (int)R.size() - 1;
            if (tail != QR_NULL) R[tail].next = ix; else head = ix;
            tail = ix;
            return ix;
        };
        auto close_until = [&](int i) {
            while (!open.empty() && open.back().last < i)
            {
                if (open.back().out != QR_NULL) { R[open.back().out].data = tail; run.local[open.back().out] = 1; }     /* last kept member */
                open.pop_back();
            }
        };
        const int n = (int)ch.size();
        for (int i = 0; i < n; )
        {
            close_until(i);
            const Node &nd = ch[i];
            if (nd.head)
            {
                if (!keep(i)) { dropped = true; i = nd.last + 1; continue; }      /* prune the array */
                open.push_back(Open{i, QR_NULL, nd.last});
                i++;
                continue;
            }
            const bool real = is_real(v.srf[nd.si]);
            if (!real || keep(i))
            {
                for (Open &o : open) if (o.out == QR_NULL) o.out = emit(ch[o.idx].e, QR_NULL);
                emit(nd.e, E[nd.e].data);
            }
            else dropped = true;
            i++;
        }
        close_until(n);
        if (!dropped && n > 0) { R.clear(); run.local.clear(); return SOURCE_LIST; }
        /* a list that keeps only a handful of surfaces does not need their bounding-volume elements (AR_ptr elements
         * only skip work, tracer.cpp:3955-4054): written flat it is a fraction of the cells.  Trnode elements stay. */
        int kept = 0;
        for (int e = head; e != QR_NULL; e = R[e].next) if (is_real(v.srf[R[e].simd])) kept++;
        if (kept <= QR_FLAT_LIST_MAX)
        {
            int nh = QR_NULL, nt = QR_NULL;
            for (int e = head; e != QR_NULL; )
            {
                const int nx = R[e].next;
                if ((R[e].kind & 3) != 1)
                {
                    if (nt != QR_NULL) R[nt].next = e; else nh = e;
                    nt = e; R[e].next = QR_NULL;
                }
                e = nx;
            }
            head = nh;
        }
        return head;
    }
};


/*
 * May a surface with bounding sphere x stand between the point lp and some point of the sphere (sc, sr)?  The hull of
 * the light position and that sphere: at parameter tau in [0,1] along the axis a sphere of radius tau sr; x meets it iff
 * min_tau |c - axis(tau)| - tau sr <= r_x; the minimum is at least d_min sqrt(1 - rho^2) - tc sr with d_min the
 * distance to the axis segment and tc its parameter (bbox_shad's role, rtgeom.cpp:1004).
 */
struct HullPred
{
    double lp[3], D[3], D2, Dl, sr, shrink; bool all;
    HullPred(const float *light_pos, const double *sc, double sr_) : sr(sr_)
    {
        for (int k = 0; k < 3; k++) { lp[k] = light_pos[k]; D[k] = sc[k] - lp[k]; }
        D2 = D[0] * D[0] + D[1] * D[1] + D[2] * D[2]; Dl = __builtin_sqrt(D2);
        const double rho = Dl > 0.0 ? sr / Dl : 2.0;
        all = !(rho < 0.95);
        shrink = all ? 0.0 : __builtin_sqrt(1.0 - rho * rho);
    }
    bool operator()(const BSphere &x, bool bounded) const
    {
        if (all || !bounded) return true;
        const double P[3] = { x.c[0] - lp[0], x.c[1] - lp[1], x.c[2] - lp[2] };
        const double t = (P[0] * D[0] + P[1] * D[1] + P[2] * D[2]) / D2;
        const double tc = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double q[3] = { P[0] - tc * D[0], P[1] - tc * D[1], P[2] - tc * D[2] };
        const double dmin = __builtin_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        const double xr = (double)x.r * 1.001 + 1e-3;
        if (t > 1.0 + (sr + xr) / Dl + 1e-3) return false;          /* behind the surface */
        return dmin * shrink <= xr + tc * sr * 1.001 + 1e-3;
    }
};

/* worker threads of the host passes: QR_HOST_THREADS, else the machine's (at most 16) */
static int host_threads()
{
    const char *te = getenv("QR_HOST_THREADS");
    int n = te ? atoi(te) : (int)std::thread::hardware_concurrency();
    if (n > 16) n = 16;
    return n < 1 ? 1 : n;
}

struct Builder
{
    const qr_scene_view &v;
    const std::vector<qr_elem> &E;          /* cells: the snapshot's, or the binning pass's                 */
    std::vector<qr_elem> *Egrow = nullptr;  /* the same vector when the compiler may append filtered chains (shadow grids) */
    std::vector<int> light_user;            /* light list head -> the one surface that uses it, -2 several, -1 none */
    int grid_min = 256;                     /* shadow lists with at least this many surfaces get a grid       */
    int dda_min = 0;                        /* lists with at least this many bounded members get a CDda (0: none) */
    uint32_t n_dda = 0;
    uint32_t n_grids = 0, n_grid_lists = 0;
    const std::vector<BSphere> &bs;
    const int cull_mode;
    std::vector<uint8_t> &blob;
    int n_srf, n_elm;
    uint32_t o_srf = 0, o_shd = 0, o_mat = 0, o_lgt = 0, o_tex = 0;

    std::vector<uint32_t> list_off;         /* surface list head -> byte offset of its program (0 unseen)   */
    std::vector<uint32_t> light_off;        /* light list head  -> byte offset                               */
    std::vector<uint8_t> list_heavy;        /* surface list head -> 1 holds a reflective, 2 a non-opaque surface */
    struct ClipKey { int head, trnode, cdef; uint32_t off; };
    std::vector<ClipKey> clip_memo;
    std::vector<int> chain_pos, chain_stamp; int stamp = 0;
    /* scratch of compile_list, kept between calls (a 1080p frame compiles thousands of short tile lists) */
    struct Tmp { int e; uint32_t op; int si; int last; bool emit; };
    std::vector<Tmp> ch;
    std::vector<int> lo_after, lo_self, open_arr, emit_idx;
    std::vector<BSphere> arr_sphere;
    QrProgramStats st = {};
    bool any_long = false;
    bool box_ok = false;                    /* every list is short: cull cells may carry boxes (QR_OPF_BOX) */
    bool any_box = false;
    float box_pad = 0.0f;

    Builder(const qr_scene_view &v_, const std::vector<qr_elem> &E_, const std::vector<BSphere> &bs_, int cm, std::vector<uint8_t> &b)
        : v(v_), E(E_), bs(bs_), cull_mode(cm), blob(b), n_srf((int)v_.hdr->n_srf), n_elm((int)E_.size())
    {
        list_off.assign((size_t)n_elm + 1, 0); light_off.assign((size_t)n_elm + 1, 0); list_heavy.assign((size_t)n_elm + 1, 0);
        chain_pos.assign((size_t)n_elm + 1, 0); chain_stamp.assign((size_t)n_elm + 1, -1);
    }

    uint32_t alloc(size_t bytes, size_t align)
    {
        size_t o = (blob.size() + align - 1) & ~(align - 1);
        if (o + bytes > 0xFFFFFFF0ull) throw Fail{QR_ERR_NOMEM, "compiled scene exceeds 4 GiB"};
        blob.resize(o + bytes, 0);
        return (uint32_t)o;
    }
    template <typename T> T *at(uint32_t off) { return (T *)(blob.data() + off); }
    uint32_t srf_off(int si) const { return o_srf + (uint32_t)si * (uint32_t)sizeof(DSurf); }

    /* ---- surface lists ---- */
    /* chains compiled so far by content (see compile_list): an open hash of entry indices, entries chained on collisions */
    struct DedupEnt { size_t first, len; uint64_t key; uint32_t off; uint8_t heavy; int32_t next; };
    std::vector<DedupEnt> dedup_ent;
    std::vector<int32_t> dedup_slot = std::vector<int32_t>(1024, -1);
    std::vector<int32_t> dedup_pool, dk;
    int32_t dedup_last = -1;                /* the entry the previous chain matched or made: neighbouring tiles repeat it */
    bool dedup_on = !(getenv("QR_LIST_DEDUP") && atoi(getenv("QR_LIST_DEDUP")) == 0);      /* QR_LIST_DEDUP=0: one program per chain (A/B) */

    bool dedup_equal(const DedupEnt &d, uint64_t key) const
    {
        return d.key == key && d.len == dk.size() && memcmp(dedup_pool.data() + d.first, dk.data(), dk.size() * sizeof(int32_t)) == 0;
    }
    void dedup_add(uint64_t key, uint32_t off, uint8_t heavy)
    {
        if ((dedup_ent.size() + 1) * 2 > dedup_slot.size())
        {
            dedup_slot.assign(dedup_slot.size() * 4, -1);
            for (size_t i = 0; i < dedup_ent.size(); i++)
            {
                int32_t &sl = dedup_slot[(size_t)(dedup_ent[i].key >> 20) & (dedup_slot.size() - 1)];
                dedup_ent[i].next = sl; sl = (int32_t)i;
            }
        }
        int32_t &sl = dedup_slot[(size_t)(key >> 20) & (dedup_slot.size() - 1)];
        dedup_ent.push_back(DedupEnt{ dedup_pool.size(), dk.size(), key, off, heavy, sl });
        sl = dedup_last = (int32_t)dedup_ent.size() - 1;
        dedup_pool.insert(dedup_pool.end(), dk.begin(), dk.end());
    }

    struct Restart {};                      /* thrown by compile_list: a long chain in a build that assumed there is none */

    uint32_t compile_list(int head)
    {
        if (head == QR_NULL) return 0;
        if (E.size() + 1 > list_off.size())
        {
            /* chains appended by the shadow grids */
            const size_t n1 = E.size() + 1;
            list_off.resize(n1, 0); light_off.resize(n1, 0); list_heavy.resize(n1, 0);
            chain_pos.resize(n1, 0); chain_stamp.resize(n1, -1);
        }
        if (list_off[head]) return list_off[head];
        stamp++;
        int n = 0;
        for (int e = head; e != QR_NULL; e = E[e].next)
        {
            if ((size_t)n >= E.size()) throw Fail{QR_ERR_ARG, "cyclic list"};
            chain_stamp[e] = stamp; chain_pos[e] = n++;
        }
        if (box_ok && n >= QR_LONG_CELLS) throw Restart{};
        /* The engine hands every screen tile a chain of its own, and most neighbours hold the same surfaces in the same order:
         * a chain whose elements (surface, kind, position of the array's last member) equal those of a chain compiled
         * before IS that program -- the cells depend on nothing else.  One copy: a third of the compile time at 1080p, and
         * the waves of a frame read a few hundred list programs through the scalar cache instead of tens of thousands. */
        uint64_t dkey = 1469598103934665603ull;
        bool dedup = dedup_on;
        dk.clear();
        for (int e = head; e != QR_NULL && dedup; e = E[e].next)
        {
            const qr_elem &el = E[e];
            int rel = -1;
            if (el.data != QR_NULL)
            {
                if (el.data < 0 || (size_t)el.data >= chain_stamp.size() || chain_stamp[el.data] != stamp) { dedup = false; break; }
                rel = chain_pos[el.data];
            }
            const int32_t w3[3] = { el.simd, el.kind, rel };
            for (int k = 0; k < 3; k++) { dk.push_back(w3[k]); dkey = (dkey ^ (uint32_t)w3[k]) * 1099511628211ull; }
        }
        if (dedup)
        {
            int32_t hit = -1;
            if (dedup_last >= 0 && dedup_equal(dedup_ent[(size_t)dedup_last], dkey)) hit = dedup_last;
            else
                for (int32_t i = dedup_slot[(size_t)(dkey >> 20) & (dedup_slot.size() - 1)]; i >= 0; i = dedup_ent[(size_t)i].next)
                    if (dedup_equal(dedup_ent[(size_t)i], dkey)) { hit = i; break; }
            if (hit >= 0)
            {
                const DedupEnt &d = dedup_ent[(size_t)hit];
                dedup_last = hit;
                list_off[head] = d.off; list_heavy[head] = d.heavy;
                return d.off;
            }
        }
        ch.clear();
        for (int e = head; e != QR_NULL; e = E[e].next) ch.push_back(Tmp{e, 0, E[e].simd, QR_NULL, true});
        lo_after.assign((size_t)n, QR_NULL); lo_self.assign((size_t)n, QR_NULL);
        /* box cull cells (QR_OPF_BOX) only in images whose lists are all short: the packet-walk kernel instance serves them */
        const bool list_boxes = box_ok;
        int local_obj = QR_NULL;
        for (int i = 0; i < n; i++)
        {
            Tmp &t = ch[i];
            const qr_elem &el = E[t.e];
            const qr_surface &s = v.srf[el.simd];
            const bool arr = s.srf_t[3] < 0, real = is_real(s);
            const bool bv = (el.kind & 3) == 1;
            uint32_t mode, type;
            bool trnode_cell = false;
            if (!arr && local_obj != QR_NULL)
            {
                mode = QR_OPF_CACHED;
                if (t.e == local_obj) local_obj = QR_NULL;
            }
            else if (s.has_trm == 0) mode = 0;
            else if (arr) { mode = 0; trnode_cell = true; local_obj = el.data; }
            else mode = QR_OPF_OWN;
            lo_self[i] = local_obj;
            const int solver = real ? s.srf_t[0] : 0;
            if (trnode_cell)
            {
                if (bv) throw Fail{QR_ERR_UNSUP, "bounding volume on a transformed array element"};
                if (el.data == QR_NULL || chain_stamp[el.data] != stamp || chain_pos[el.data] < i)
                    throw Fail{QR_ERR_ARG, "trnode's last element is not behind it in its list"};
                type = QR_OPT_TRNODE; t.last = chain_pos[el.data];
            }
            else if (bv)
            {
                if (el.data == QR_NULL || chain_stamp[el.data] != stamp || chain_pos[el.data] < i)
                    throw Fail{QR_ERR_ARG, "array's last element is not behind it in its list"};
                type = QR_OPT_BV; t.last = chain_pos[el.data];
            }
            else if (solver == 1) type = QR_OPT_PLANE;
            else if (solver == 2) type = QR_OPT_QUADRIC;
            else if (solver == 3) type = QR_OPT_TWOPLANE;
            else { type = 0; t.emit = false; }          /* marker: touches only state nobody reads */
            if (t.emit && type != QR_OPT_TRNODE && ((s.shift != 0) != (mode != 0)))
                throw Fail{QR_ERR_UNSUP, "surface reads the trnode-space diff outside a trnode (or the reverse)"};
            uint32_t op = type | mode;
            if (s.has_trm != 1) op |= QR_OPF_FULLM;
            if (type & QR_OPT_SOLVER)
            {
                const uint32_t ak = (s.axes >> 4) & 3u, ai = (s.axes >> 0) & 3u;
                if (ak == 0) op |= QR_OPF_KX; else if (ak == 1) op |= QR_OPF_KY;
                if ((s.axes >> 10) & 1u) op |= QR_OPF_SGNK;
                if (ai == 0) op |= QR_OPF_IX; else if (ai == 1) op |= QR_OPF_IY;
                auto no_shadow = [](int p) { return (p & QR_PROP_LIGHT) || ((p & QR_PROP_TRANSP) && !(p & QR_PROP_REFRACT)); };
                const bool n0 = no_shadow(s.props[0]), n1 = no_shadow(s.props[1]);
                if (n0 && n1) op |= QR_OPF_NOSHAD; else if (n0 || n1) op |= QR_OPF_SIDESHAD;
                if (s.clip != QR_NULL) op |= QR_OPF_CLIP;
                if (s.conic != 0) op |= QR_OPF_CONIC;
                const bool open_shape = s.srf_t[0] == 1 || !(s.sci[0] > 0.0f && s.sci[1] > 0.0f && s.sci[2] > 0.0f);
                const bool want = cull_mode >= 3 || (cull_mode == 2 && open_shape) || (cull_mode == 1 && s.srf_t[0] == 1);
                if (want && bs[el.simd].r < 1e18f)
                {
                    op |= QR_OPF_CULL;
                    /* box or sphere: whichever shows the smaller silhouette on average (a convex body's mean projected area
                     * is a quarter of its surface) */
                    const BSphere &bb = bs[el.simd];
                    if (list_boxes && bb.lo[0] <= bb.hi[0])
                    {
                        const double a = (double)bb.hi[0] - bb.lo[0], b = (double)bb.hi[1] - bb.lo[1], c = (double)bb.hi[2] - bb.lo[2];
                        if (0.5 * (a * b + b * c + c * a) < 0.8 * 3.14159265358979 * (double)bb.r * bb.r) { op |= QR_OPF_BOX; any_box = true; }
                    }
                }
            }
            t.op = op;
            lo_after[i] = local_obj;
        }
        /* static state must not depend on whether a ray walked through an array or skipped it (AR_skp:
         * e = last; if (e == local_obj) local_obj = NULL), and arrays must nest */
        {
            std::vector<int> &open = open_arr;
            open.clear();
            for (int i = 0; i < n; i++)
            {
                while (!open.empty() && open.back() < i) open.pop_back();
                if (!(ch[i].op & QR_OPT_BV) || !ch[i].emit) continue;
                const int j = ch[i].last;
                int lo = lo_self[i];
                if (lo == ch[j].e) lo = QR_NULL;
                if (lo != lo_after[j]) throw Fail{QR_ERR_UNSUP, "a trnode's range crosses the end of a bounding-volume array"};
                if (!open.empty() && j > open.back()) throw Fail{QR_ERR_UNSUP, "bounding-volume arrays are not nested"};
                open.push_back(j);
            }
        }
        /* bounding-volume arrays get a conservative sphere of their own (ours): the union of the members' bounding
         * spheres around the first bounded member's centre.  The walk skips an array that lies entirely behind a
         * ray's origin or entirely beyond its current depth bound -- the reference's own test only asks whether the
         * LINE meets the volume.  No sphere (no cull) when a member is unbounded. */
        arr_sphere.resize((size_t)n);
        if (cull_mode >= 3)
            for (int i = 0; i < n; i++)
            {
                if (!(ch[i].op & QR_OPT_BV) || !ch[i].emit) continue;
                bool ok = true; int first = -1;
                for (int k = i + 1; k <= ch[i].last && ok; k++)
                    if (ch[k].emit && (ch[k].op & QR_OPT_SOLVER))
                    {
                        if (!(bs[ch[k].si].r < 1e18f)) ok = false;
                        else if (first < 0) first = k;
                    }
                if (!ok || first < 0) continue;
                const BSphere &c0 = bs[ch[first].si];
                double R = 0.0;
                for (int k = i + 1; k <= ch[i].last; k++)
                    if (ch[k].emit && (ch[k].op & QR_OPT_SOLVER))
                    {
                        const BSphere &m = bs[ch[k].si];
                        const double dx = (double)m.c[0] - c0.c[0], dy = (double)m.c[1] - c0.c[1], dz = (double)m.c[2] - c0.c[2];
                        const double d = __builtin_sqrt(dx * dx + dy * dy + dz * dz) + (double)m.r;
                        if (d > R) R = d;
                    }
                R = R * 1.0001 + 1e-4;
                if (!(R < 1e18)) continue;
                arr_sphere[i].c[0] = c0.c[0]; arr_sphere[i].c[1] = c0.c[1]; arr_sphere[i].c[2] = c0.c[2]; arr_sphere[i].r = (float)R * 1.0001f;
                ch[i].op |= QR_OPF_CULL;
                /* the volume itself a plain sphere around all members' bounds?  Then it is the cull sphere. */
                const qr_surface &q = v.srf[ch[i].si];
                if (!(ch[i].op & QR_OPF_LOCAL) && q.has_trm == 0 && q.sci[0] == 1.0f && q.sci[1] == 1.0f && q.sci[2] == 1.0f
                    && q.sci[3] > 0.0f && q.sci[3] < 1e30f)
                {
                    const double rq = __builtin_sqrt((double)q.sci[3]);
                    bool inside = true;
                    for (int k = i + 1; k <= ch[i].last && inside; k++)
                        if (ch[k].emit && (ch[k].op & QR_OPT_SOLVER))
                        {
                            const BSphere &m = bs[ch[k].si];
                            const double dx = (double)m.c[0] - q.pos[0], dy = (double)m.c[1] - q.pos[1], dz = (double)m.c[2] - q.pos[2];
                            if (__builtin_sqrt(dx * dx + dy * dy + dz * dz) + (double)m.r > rq * 1.0001) inside = false;
                        }
                    if (inside)
                    {
                        arr_sphere[i].c[0] = q.pos[0]; arr_sphere[i].c[1] = q.pos[1]; arr_sphere[i].c[2] = q.pos[2];
                        arr_sphere[i].r = (float)(rq * 1.0002 + 1e-4);
                        ch[i].op |= QR_OPF_SPHBV;
                    }
                }
            }
        bool has_trnode = false;
        for (int i = 0; i < n; i++)
            if (ch[i].emit && ((ch[i].op & QR_OPT_TRNODE) || (ch[i].op & QR_OPF_CACHED))) has_trnode = true;
        /* emit */
        emit_idx.assign((size_t)n + 1, 0);
        int ne = 0;
        int n_emitted = 0;
        /* slot index of every cell: a bounding-volume cell takes two slots (its extension carries the volume) */
        for (int i = 0; i < n; i++) { emit_idx[i] = ne; if (ch[i].emit) { ne += (ch[i].op & QR_OPT_BV) ? 2 : 1; n_emitted++; } }
        emit_idx[n] = ne;
        /* uniform grid (CDda) for long world-space lists without clipper programs: decided before the cells are placed,
         * the record sits in the 64 bytes in front of the program */
        bool want_dda = false;
        if (dda_min > 0 && !has_trnode)
        {
            int n_small = 0; bool ok = true;
            for (int i = 0; i < n && ok; i++)
            {
                if (!ch[i].emit) continue;
                if (ch[i].op & (QR_OPF_LOCAL | QR_OPF_CLIP)) ok = false;
                if ((ch[i].op & QR_OPT_SOLVER) && bs[ch[i].si].r < 1e17f) n_small++;
            }
            want_dda = ok && n_small >= dda_min;
        }
        /* Lists the per-lane walk with hand-over may take (world-space cells only, no clipper program) carry their LENGTH in the
         * word in front of the program -- in a cell of their own, or in the grid record's spare word: (bytes of cells in front of
         * the END cell) | 1 when no cell is a bounding volume (every multiple of 32 bytes is then a cell boundary).  walk_pool
         * cuts such a flat list in halves for idle lanes (qr_walk.hpp); a list program ends with an END cell, so without this
         * word a lane knows where its range ends only when it gets there. */
        bool pre_world = true, pre_div = true; int pre_bv = 0;
        for (int i = 0; i < n; i++)
        {
            if (!ch[i].emit) continue;
            if ((ch[i].op & QR_OPT_TRNODE) || (ch[i].op & QR_OPF_LOCAL)) pre_world = false;
            if (ch[i].op & QR_OPF_CLIP) pre_div = false;
            if (ch[i].op & QR_OPT_BV) pre_bv++;
        }
        const bool want_len = pre_world && pre_div;
        const size_t front = want_dda ? sizeof(CDda) : (want_len ? sizeof(CCell) : 0);
        const uint32_t base = alloc((size_t)(ne + 1) * sizeof(CCell) + front, 64);
        const uint32_t off = base + (uint32_t)front;
        if (want_len) *at<uint32_t>(off - 4u) = (uint32_t)ne * (uint32_t)sizeof(CCell) | (pre_bv == 0 ? 1u : 0u);
        list_off[head] = off;
        for (int i = 0; i < n; i++)
        {
            if (!ch[i].emit) continue;
            CCell c;
            memset(&c, 0, sizeof(c));
            c.op = ch[i].op; c.srf = srf_off(ch[i].si);
            if (c.op & QR_OPT_BV) c.end = off + (uint32_t)emit_idx[ch[i].last + 1] * (uint32_t)sizeof(CCell);
            c.r = __builtin_inff();
            if (!(c.op & QR_OPT_BV)) { c.r2 = __builtin_inff(); c.r2x = __builtin_inff(); }   /* no cull: no test of walk_pool's fails */
            if (c.op & QR_OPF_BOX)
            {
                /* + box_pad: the slab test's own rounding, 2e-6 of the scene's largest coordinate (qr_walk.hpp) */
                const BSphere &bb = bs[ch[i].si];
                c.r2 = bb.lo[0] - box_pad; c.r2x = bb.lo[1] - box_pad; c.cx = bb.lo[2] - box_pad;
                c.cy = bb.hi[0] + box_pad; c.cz = bb.hi[1] + box_pad; c.r = bb.hi[2] + box_pad;
            }
            else if (c.op & QR_OPF_CULL)
            {
                const BSphere &bsp = (c.op & QR_OPT_BV) ? arr_sphere[i] : bs[ch[i].si];
                c.cx = bsp.c[0]; c.cy = bsp.c[1]; c.cz = bsp.c[2]; c.r = bsp.r;
                if (!(c.op & QR_OPT_BV)) { c.r2 = bsp.r * bsp.r; c.r2x = c.r2 * 1.01f; }      /* BV: the slot holds `end` */
            }
            if (c.op & QR_OPT_BV) c.r2x = (c.op & QR_OPF_SPHBV) ? -1.0f : __builtin_inff();
            *at<CCell>(off + (uint32_t)emit_idx[i] * (uint32_t)sizeof(CCell)) = c;
            if (c.op & QR_OPT_BV)
            {
                /* extension slot: what AR_ptr reads of the volume's surface record, so that the walk needs no
                 * second, dependent load for it */
                const qr_surface &q = v.srf[ch[i].si];
                CBvExt x;
                memset(&x, 0, sizeof(x));
                for (int k = 0; k < 3; k++) x.pos[k] = q.pos[k];
                for (int k = 0; k < 4; k++) x.sci[k] = q.sci[k];
                /* hand-over boundary: the child (direct member or nested array) that starts nearest to the middle of the
                 * array's cells.  Only where a walk carries no trnode state and the array is worth splitting. */
                const int s0 = emit_idx[i] + 2, s1 = emit_idx[ch[i].last + 1];
                if (!has_trnode && s1 - s0 >= 12)
                {
                    int best = -1;
                    for (int k = i + 1; k <= ch[i].last; )
                    {
                        if (ch[k].emit && emit_idx[k] > s0 && (best < 0 || abs(2 * emit_idx[k] - (s0 + s1)) < abs(2 * emit_idx[best] - (s0 + s1))))
                            best = k;
                        k = (ch[k].emit && (ch[k].op & QR_OPT_BV)) ? ch[k].last + 1 : k + 1;
                    }
                    if (best >= 0) x.mid = off + (uint32_t)emit_idx[best] * (uint32_t)sizeof(CCell);
                }
                *at<CBvExt>(off + (uint32_t)(emit_idx[i] + 1) * (uint32_t)sizeof(CCell)) = x;
            }
        }
        /* END cell: already zero */
        st.n_lists++; st.n_cells += (uint32_t)n_emitted; st.n_dropped += (uint32_t)(n - n_emitted);
        /* flags in the offset's low bits: may the per-lane walk take this list, is it a long hierarchy */
        uint32_t lf = QR_LISTF_DIV;
        int n_bv = 0;
        uint8_t heavy = 0;
        for (int i = 0; i < n; i++)
        {
            const qr_surface &q = v.srf[ch[i].si];
            if (is_real(q))
                for (int k = 0; k < 2; k++)
                {
                    if (q.props[k] & QR_PROP_REFLECT) heavy |= 1;
                    if (!(q.props[k] & QR_PROP_OPAQUE)) heavy |= 2;
                }
            if (!ch[i].emit) continue;
            if (ch[i].op & QR_OPF_CLIP) lf &= ~QR_LISTF_DIV;
            if (ch[i].op & QR_OPT_BV) n_bv++;
        }
        /* a long hierarchy: rays part ways on it.  (96 cells until round 3: demo scene 2's ~110-cell lists then put its frames on
         * the per-lane kernel instance, which renders them at 3 waves per SIMD -- 34.8 against 46.5 Grays/s with this one.) */
        if (n_emitted >= QR_LONG_CELLS && n_bv >= 4) { lf |= QR_LISTF_LONG; any_long = true; }
        bool world = true;
        for (int i = 0; i < n; i++)
            if (ch[i].emit && ((ch[i].op & QR_OPT_TRNODE) || (ch[i].op & QR_OPF_LOCAL))) world = false;
        if (world) lf |= QR_LISTF_WORLD;
        if (want_dda && world && (lf & QR_LISTF_DIV)) { build_dda(base, off, n); lf |= QR_LISTF_DDA; any_long = true; }
        list_off[head] = off | lf;
        list_heavy[head] = heavy;
        if (dedup) dedup_add(dkey, off | lf, heavy);
        return off | lf;
    }

    /*
     * CDda of the list just emitted at `off` (ch / emit_idx still describe it): bins the members' bounding spheres.
     */
    void build_dda(uint32_t rec_off, uint32_t off, int n)
    {
        /* members: emitted solver cells; small ones span the grid */
        double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
        std::vector<int> mem;
        std::vector<float> rad;
        for (int i = 0; i < n; i++)
            if (ch[i].emit && (ch[i].op & QR_OPT_SOLVER)) { mem.push_back(i); if (bs[ch[i].si].r < 1e17f) rad.push_back(bs[ch[i].si].r); }
        std::vector<float> sorted = rad;
        std::sort(sorted.begin(), sorted.end());
        const double r_big = 4.0 * (double)sorted[sorted.size() / 2] + 1e-3;        /* larger than 4 median radii: tested up front */
        int n_small = 0;
        for (int i : mem)
        {
            const BSphere &b = bs[ch[i].si];
            if (!(b.r < r_big)) continue;
            n_small++;
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)b.c[k] - b.r); hi[k] = std::max(hi[k], (double)b.c[k] + b.r); }
        }
        CDda g;
        memset(&g, 0, sizeof(g));
        double ext[3], vol = 1.0;
        for (int k = 0; k < 3; k++) { lo[k] -= 1e-3; hi[k] += 1e-3; ext[k] = hi[k] - lo[k]; vol *= ext[k]; }
        const char *pm = getenv("QR_DDA_CELLS");
        const double per_member = pm ? atof(pm) : 2.0;          /* cells per member: 0.5 .. 8 are within 5 % of each other on the 10k scene */
        const double c0 = __builtin_cbrt(vol / ((per_member > 0.01 ? per_member : 2.0) * (n_small > 0 ? n_small : 1)));
        int dim[3];
        for (int k = 0; k < 3; k++)
        {
            int d = (int)(ext[k] / c0 + 0.5);
            dim[k] = d < 1 ? 1 : (d > (int)QR_GRID_MAX ? (int)QR_GRID_MAX : d);
            g.org[k] = (float)lo[k]; g.size[k] = (float)(ext[k] / dim[k]); g.inv[k] = (float)(dim[k] / ext[k]);
        }
        g.dims = (uint32_t)dim[0] | ((uint32_t)dim[1] << 8) | ((uint32_t)dim[2] << 16);
        const size_t n_cells = (size_t)dim[0] * dim[1] * dim[2];
        /* count, then fill (members in list order inside a cell) */
        auto range = [&](const BSphere &b, int k, int &a0, int &a1) {
            /* cells overlap: members are entered a thousandth of a cell beyond their sphere's box, so that a hit within
             * rounding of a cell face is found from either side */
            const double pad = 1e-3 * g.size[k] + 1e-4;
            a0 = (int)__builtin_floor(((double)b.c[k] - b.r - pad - lo[k]) * dim[k] / ext[k]);
            a1 = (int)__builtin_floor(((double)b.c[k] + b.r + pad - lo[k]) * dim[k] / ext[k]);
            a0 = a0 < 0 ? 0 : a0; a1 = a1 >= dim[k] ? dim[k] - 1 : a1;
        };
        std::vector<uint32_t> cnt(n_cells + 1, 0);
        std::vector<int> outl;
        for (int i : mem)
        {
            const BSphere &b = bs[ch[i].si];
            if (!(b.r < r_big)) { outl.push_back(i); continue; }
            int x0, x1, y0, y1, z0, z1;
            range(b, 0, x0, x1); range(b, 1, y0, y1); range(b, 2, z0, z1);
            for (int z = z0; z <= z1; z++) for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++)
                cnt[((size_t)z * dim[1] + y) * dim[0] + x + 1]++;
        }
        g.n_out = (uint32_t)outl.size();
        cnt[0] = g.n_out;
        for (size_t c = 1; c <= n_cells; c++) cnt[c] += cnt[c - 1];
        const uint32_t n_refs = cnt[n_cells];
        g.n_refs = n_refs;
        std::vector<CCell> refs(n_refs);
        auto ref_of = [&](int i) {
            CCell c = *at<CCell>(off + (uint32_t)emit_idx[i] * (uint32_t)sizeof(CCell));
            uint32_t pos = off + (uint32_t)emit_idx[i] * (uint32_t)sizeof(CCell);
            memcpy(&c.r2x, &pos, 4);            /* the original cell: list order decides between equal depths */
            return c;
        };
        for (size_t q = 0; q < outl.size(); q++) refs[q] = ref_of(outl[q]);
        std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
        for (int i : mem)
        {
            const BSphere &b = bs[ch[i].si];
            if (!(b.r < r_big)) continue;
            int x0, x1, y0, y1, z0, z1;
            range(b, 0, x0, x1); range(b, 1, y0, y1); range(b, 2, z0, z1);
            const CCell c = ref_of(i);
            for (int z = z0; z <= z1; z++) for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++)
                refs[fill[((size_t)z * dim[1] + y) * dim[0] + x]++] = c;
        }
        g.cells = alloc((n_cells + 1) * 4, 16);
        memcpy(at<uint32_t>(g.cells), cnt.data(), (n_cells + 1) * 4);
        g.refs = alloc((size_t)(n_refs + 2) * sizeof(CCell), 32);        /* + slack: the walk loads 32 bytes at its cursor */
        if (n_refs) memcpy(at<CCell>(g.refs), refs.data(), (size_t)n_refs * sizeof(CCell));
        g.pad[1] = *at<uint32_t>(off - 4u);                            /* the list's length word (compile_list) lives in the record's last word */
        *at<CDda>(rec_off) = g;
        n_dda++;
    }

    /* ---- clipper programs ---- */
    uint32_t compile_clip(int owner)
    {
        const qr_surface &s = v.srf[owner];
        if (s.clip == QR_NULL) return 0;
        const int cdef = s.c_def != 0 ? 1 : 0;
        for (const ClipKey &k : clip_memo)
            if (k.head == s.clip && k.trnode == s.trnode && k.cdef == cdef) return k.off;
        std::vector<CClip> prog;
        int redx = QR_NULL;
        int group_last = -1;                /* index in prog of the last cell emitted for the open cached group */
        for (int e = s.clip; e != QR_NULL; e = E[e].next)
        {
            const qr_elem &el = E[e];
            /* the cached trnode space of a group ends behind its last element (redx), whether or not that element emits a cell */
            struct Closer { int &redx, &group_last; std::vector<CClip> &prog; int e;
                            ~Closer() { if (group_last >= 0 && redx == QR_NULL) { prog[(size_t)group_last].op |= QR_CLF_LASTC; group_last = -1; } } } closer{redx, group_last, prog, e};
            CClip c;
            memset(&c, 0, sizeof(c));
            if (el.simd == QR_NULL)
            {
                c.op = el.data > 0 ? QR_CLT_LEAVE : (QR_CLT_ENTER | (cdef ? QR_CLF_CDEF : 0u));
                prog.push_back(c);
                if (redx != QR_NULL) group_last = (int)prog.size() - 1;
                continue;
            }
            const qr_surface &k = v.srf[el.simd];
            const bool karr = k.srf_t[3] < 0;
            c.srf = srf_off(el.simd);
            if (k.has_trm != 1) c.op |= QR_CLF_FULLM;
            uint32_t mode;
            if (!karr)
            {
                if (redx != QR_NULL) { mode = QR_CLF_CACHED; if (e == redx) redx = QR_NULL; }
                else mode = k.has_trm != 0 ? QR_CLF_OWN : 0u;
            }
            else if (el.simd == s.trnode)
            {
                if (s.has_trm == 0) throw Fail{QR_ERR_UNSUP, "clipper trnode shared with an untransformed surface"};
                c.op |= QR_CLT_TRSAME; redx = el.data; prog.push_back(c); group_last = (int)prog.size() - 1; if (e == redx) redx = QR_NULL; continue;
            }
            else
            {
                if (k.has_trm == 0)
                {
                    /* array without transform: outside a cached trnode it writes only state nobody reads */
                    if (redx != QR_NULL) throw Fail{QR_ERR_UNSUP, "untransformed array inside a clipper trnode"};
                    continue;
                }
                c.op |= QR_CLT_TRNODE; redx = el.data; prog.push_back(c); group_last = (int)prog.size() - 1; if (e == redx) redx = QR_NULL; continue;
            }
            const int ckind = k.srf_t[2];
            if (ckind == 0) continue;                   /* no clip function: no effect */
            if ((k.shift != 0) != (mode != 0))
                throw Fail{QR_ERR_UNSUP, "clipper reads the trnode-space hit outside a trnode (or the reverse)"};
            c.op |= (ckind == 1 ? QR_CLT_PLANE : ckind == 2 ? QR_CLT_QUADJ : QR_CLT_QUAD) | mode;
            if (el.data < 0) c.op |= QR_CLF_INNER;
            const uint32_t ak = (k.axes >> 4) & 3u;
            if (ak == 0) c.op |= QR_CLF_KX; else if (ak == 1) c.op |= QR_CLF_KY;
            if ((k.axes >> 10) & 1u) c.op |= QR_CLF_SGNK;
            if (ckind == 1 && mode != QR_CLF_OWN && ak <= 2)
            {
                /* fast plane cell: the plane's position along its axis with the sign folded in (qr_program.h) */
                c.op |= QR_CLF_FASTPL;
                const float val = (c.op & QR_CLF_SGNK) ? -k.pos[ak] : k.pos[ak];
                memcpy(&c.aux, &val, 4);
                c.sgn = (c.op & QR_CLF_SGNK) ? 0x80000000u : 0u;
                c.mx = ak == 0 ? 0xFFFFFFFFu : 0u; c.my = ak == 1 ? 0xFFFFFFFFu : 0u; c.mz = ak == 2 ? 0xFFFFFFFFu : 0u;
            }
            prog.push_back(c);
            if (mode == QR_CLF_CACHED) group_last = (int)prog.size() - 1;
        }
        CClip endc;
        memset(&endc, 0, sizeof(endc));
        prog.push_back(endc);
        const uint32_t off = alloc(prog.size() * sizeof(CClip), 32);
        memcpy(at<CClip>(off), prog.data(), prog.size() * sizeof(CClip));
        clip_memo.push_back(ClipKey{s.clip, s.trnode, cdef, off});
        st.n_clip_cells += (uint32_t)prog.size() - 1;
        return off;
    }

    /* ---- light lists ---- */
    /*
     * Shadow lists by hit position for surface `si` (an untransformed plane with a finite clip rectangle) and light `lg`,
     * whose shadow list `sh` holds many surfaces: CGrid, qr_program.h.  Returns the record's offset | QR_LISTF_GRID, or 0.
     */
    uint32_t compile_grid(int si, int lg, int sh)
    {
        const qr_surface &s = v.srf[si];
        if (!is_real(s) || s.srf_t[0] != 1 || s.has_trm != 0 || s.shift != 0) return 0;
        const int k = (int)((s.axes >> 4) & 3);
        if (k > 2) return 0;
        const int a = k == 0 ? 1 : 0, b = k == 2 ? 1 : 2;
        const bool fin = (s.minmax_t & (1u << a)) && (s.minmax_t & (1u << (3 + a))) && (s.minmax_t & (1u << b)) && (s.minmax_t & (1u << (3 + b)));
        if (!fin) return 0;
        const double lo_a = s.min[a], hi_a = s.max[a], lo_b = s.min[b], hi_b = s.max[b];
        const double ea = hi_a - lo_a, eb = hi_b - lo_b;
        if (!(ea > 1e-3) || !(eb > 1e-3) || !(ea < 1e18) || !(eb < 1e18)) return 0;
        /* long enough to be worth it? */
        int n_real = 0;
        for (int e = sh; e != QR_NULL && n_real < grid_min; e = E[e].next) if (is_real(v.srf[E[e].simd])) n_real++;
        if (n_real < grid_min) return 0;
        ListFilter lf(v, bs, *Egrow, sh);
        const double cell = (ea > eb ? ea : eb) / (double)QR_GRID_MAX;
        auto cells = [&](double ext) { int n = (int)(ext / cell + 0.5); return n < 1 ? 1 : (n > (int)QR_GRID_MAX ? (int)QR_GRID_MAX : n); };
        const int nx = cells(ea), ny = cells(eb);
        CGrid g;
        g.nx = (uint32_t)nx; g.ny = (uint32_t)ny; g.comps = (uint32_t)a | ((uint32_t)b << 2);
        g.org_a = (float)lo_a; g.org_b = (float)lo_b;
        g.inv_a = (float)(nx / ea); g.inv_b = (float)(ny / eb);
        std::vector<uint32_t> table((size_t)nx * ny, 0u);
        const double ca = ea / nx, cb = eb / ny;
        /* cells overlap: the kernel's cell index comes from the fp32 local hit (error far below a thousandth of a cell
         * at any sane scale), and the shadow ray starts at the world-space hit, the same point up to rounding */
        const double pad_a = 2e-3 * ca + 1e-3, pad_b = 2e-3 * cb + 1e-3;
        const double pr = __builtin_sqrt((0.5 * ca + pad_a) * (0.5 * ca + pad_a) + (0.5 * cb + pad_b) * (0.5 * cb + pad_b)) + 1e-3;
        /* the chains of the cells: filtered by worker threads (a cell's filter reads the chain and nothing else: 16 us each,
         * 4 x 4 096 of them are 0.27 s of config 5's upload on one core), then appended to E and compiled in cell order -- the
         * image a single thread builds.  QR_HOST_THREADS=1: one thread. */
        const int n_cells = nx * ny;
        std::vector<ListFilter::Run> runs((size_t)n_cells);
        std::vector<int> heads((size_t)n_cells, QR_NULL);
        auto filter_cells = [&](int c0, int c1) {
            for (int c = c0; c < c1; c++)
            {
                const int i = c % nx, j = c / nx;
                double pc[3] = { s.pos[0], s.pos[1], s.pos[2] };
                pc[a] += lo_a + (i + 0.5) * ca; pc[b] += lo_b + (j + 0.5) * cb;
                const HullPred pred(v.lgt[lg].pos, pc, pr);
                heads[(size_t)c] = lf.filter_run(pred, runs[(size_t)c]);
            }
        };
        int n_thr = host_threads();
        if (n_thr > n_cells / 64) n_thr = n_cells / 64;
        if (n_thr <= 1) filter_cells(0, n_cells);
        else
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < n_thr; t++)
                pool.emplace_back(filter_cells, (int)((long long)n_cells * t / n_thr), (int)((long long)n_cells * (t + 1) / n_thr));
            for (std::thread &t : pool) t.join();
        }
        for (int c = 0; c < n_cells; c++)
        {
            const int h = lf.append(runs[(size_t)c], heads[(size_t)c]);
            runs[(size_t)c] = ListFilter::Run();
            table[(size_t)c] = compile_list(h);
            n_grid_lists++;
        }
        g.table = alloc(table.size() * 4, 4);
        memcpy(at<uint32_t>(g.table), table.data(), table.size() * 4);
        const uint32_t off = alloc(sizeof(CGrid), 32);
        *at<CGrid>(off) = g;
        n_grids++;
        any_long = true;                /* the kernel instance that knows grids */
        return off | QR_LISTF_GRID;
    }

    uint32_t compile_lights(int head)
    {
        if (head == QR_NULL) return 0;
        if (light_off[head]) return light_off[head];
        std::vector<CLight> ls;
        for (int e = head; e != QR_NULL; e = E[e].next)
        {
            CLight l;
            l.lgt = o_lgt + (uint32_t)E[e].simd * (uint32_t)sizeof(qr_light);
            const int sh = E[e].data, lg = E[e].simd;       /* copies: compile_grid appends to E */
            l.shadow = compile_list(sh);
            if (Egrow != nullptr && light_user[head] >= 0 && sh != QR_NULL)
                if (const uint32_t g = compile_grid(light_user[head], lg, sh)) l.shadow = g;
            ls.push_back(l);
        }
        ls.back().lgt |= QR_CLIGHT_LAST;
        const uint32_t off = alloc(ls.size() * sizeof(CLight), 8);
        memcpy(at<CLight>(off), ls.data(), ls.size() * sizeof(CLight));
        light_off[head] = off;
        return off;
    }
};

} // namespace

/* one attempt.  assume_short: take every chain for shorter than QR_LONG_CELLS without measuring them first (the engine's scenes:
 * tens of thousands of tile chains of two or three elements); compile_list throws Restart at the first chain that is not */
static int program_build_once(const qr_scene_view &v, const std::vector<qr_elem> &E, const std::vector<int32_t> &T,
                              const qr_frame &frm, const std::vector<BSphere> &bs, QrProgram &out, std::string &err, int sched_blocks,
                              bool verify, bool assume_short, bool &restart)
{
    static const bool timing = getenv("QR_COMPILE_TIMING") != nullptr;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto tick = [&](const char *what) {
        if (!timing) return;
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "  compile %-10s %.3f ms\n", what, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
        ts0 = t1;
    };
    const int n_srf = (int)v.hdr->n_srf, n_mat = (int)v.hdr->n_mat, n_lgt = (int)v.hdr->n_lgt, n_tex = (int)v.hdr->n_texels;
    const char *cm = getenv("QR_CULL");                 /* 0 off, 1 planes, 2 planes + open quadrics, 3 all */
    const int cull_mode = cm ? atoi(cm) : 3;
    out.blob.clear();
    out.frm = frm;
    if ((size_t)frm.tls_row * frm.tls_col != T.size()) { err = "tile grid does not match the tile array"; return QR_ERR_ARG; }
    try
    {
        out.blob.reserve(sizeof(DevHeader) + (size_t)(n_srf + 1) * (sizeof(DSurf) + sizeof(DShade)) + (size_t)(n_mat + 1) * sizeof(qr_material)
                         + (size_t)n_tex * 4 + T.size() * 4 + E.size() * 80 + (size_t)frm.frm_w * frm.frm_h / 8 + 65536);
        /* shadow grids (CGrid) append filtered chains to the cell array: work on a copy then.  Only scenes with long
         * shadow lists on clipped planes get any, so the per-frame compile of the engine's scenes never copies. */
        const char *ge = getenv("QR_GRID");
        const int grid_min = ge ? atoi(ge) : 256;           /* 0 turns the grids off */
        std::vector<qr_elem> Eg;
        bool want_grids = false;
        if (grid_min > 0 && E.size() >= (size_t)grid_min)
            for (int i = 0; i < n_srf && !want_grids; i++)
            {
                const qr_surface &q = v.srf[i];
                if (is_real(q) && q.srf_t[0] == 1 && q.has_trm == 0 && q.shift == 0 && (q.lst[0] != QR_NULL || q.lst[2] != QR_NULL)) want_grids = true;
            }
        if (want_grids) Eg = E;
        tick("pre");
        Builder b(v, want_grids ? Eg : E, bs, cull_mode, out.blob);
        tick("builder");
        {
            const char *de = getenv("QR_DDA");          /* 0 turns the uniform grids off */
            b.dda_min = de ? atoi(de) : 512;
        }
        if (want_grids)
        {
            b.Egrow = &Eg; b.grid_min = grid_min;
            b.light_user.assign(E.size() + 1, -1);
            for (int i = 0; i < n_srf; i++)
            {
                const qr_surface &q = v.srf[i];
                if (!is_real(q)) continue;
                for (int k = 0; k < 2; k++)
                {
                    const int h = q.lst[k * 2];
                    if (h == QR_NULL) continue;
                    b.light_user[h] = (b.light_user[h] == -1 || b.light_user[h] == i) ? i : -2;
                }
            }
        }
        /* fixed sections; index n_* is a zero record so that masked-off lanes may read it */
        b.alloc(sizeof(DevHeader), 256);
        b.o_srf = b.alloc((size_t)(n_srf + 1) * sizeof(DSurf), 128);
        if (b.o_srf != QR_OFF_SRF) throw Fail{QR_ERR_ARG, "layout: DSurf array is not at QR_OFF_SRF"};
        b.o_shd = b.alloc((size_t)(n_srf + 1) * sizeof(DShade), 32);
        b.o_mat = b.alloc((size_t)(n_mat + 1) * sizeof(qr_material), 128);
        b.o_lgt = b.alloc((size_t)(n_lgt + 1) * sizeof(qr_light), 64);
        b.o_tex = b.alloc((size_t)(n_tex + 1) * 4, 16);
        const uint32_t o_til = b.alloc((T.size() + 1) * 4, 16);

        /* wave schedule geometry */
        const int fw = frm.fsaa == 2 ? 4 : 8, fh = frm.fsaa == 0 ? 8 : 4;
        const int nbx = (frm.frm_w + fw - 1) / fw, nby = (frm.frm_h + fh - 1) / fh;
        if (nbx > 0x3FFF || nby > 0x3FFF) throw Fail{QR_ERR_ARG, "frame too large"};
        const size_t n_sched = (size_t)nbx * nby;
        const uint32_t o_ord = b.alloc(n_sched * 8 + 16, 16);

        /* materials, lights, texels */
        memcpy(b.at<qr_material>(b.o_mat), v.mat, (size_t)n_mat * sizeof(qr_material));
        for (int i = 0; i < n_mat; i++) b.at<qr_material>(b.o_mat)[i].tex = (int32_t)(b.o_tex + (uint32_t)v.mat[i].tex * 4u);
        b.at<qr_material>(b.o_mat)[n_mat].tex = (int32_t)b.o_tex;
        b.at<qr_material>(b.o_mat)[n_mat].clamp = 1.0f;
        memcpy(b.at<qr_light>(b.o_lgt), v.lgt, (size_t)n_lgt * sizeof(qr_light));
        memcpy(b.at<uint32_t>(b.o_tex), v.texels, (size_t)n_tex * 4);
        tick("fixed");

        /* box cull cells (QR_OPF_BOX) for images whose lists are all short: no list of such an image can be flagged as a long
         * hierarchy or get a uniform / shadow grid (each needs a chain of >= QR_LONG_CELLS elements), so only packet walks read its cells.
         * QR_BOX=0: spheres only (A/B runs, tests). */
        {
            const char *be = getenv("QR_BOX");
            const bool box_lists = cull_mode >= 3 && !(be && atoi(be) == 0);
            {
                double big = 0.0;
                for (int k = 0; k < 3; k++) big = std::max(big, (double)__builtin_fabsf(frm.org[k]));
                for (int i = 0; i < n_srf; i++)
                    if (bs[i].lo[0] <= bs[i].hi[0])
                        for (int k = 0; k < 3; k++) big = std::max(big, std::max((double)__builtin_fabsf(bs[i].lo[k]), (double)__builtin_fabsf(bs[i].hi[k])));
                b.box_pad = (float)(2e-6 * big);
            }
            b.box_ok = box_lists && assume_short;       /* the second attempt: compile_list met a chain of QR_LONG_CELLS elements */
        }
        tick("setup");
        /* tile lists */
        std::vector<uint32_t> tile_off(T.size());
        for (size_t i = 0; i < T.size(); i++) tile_off[i] = b.compile_list(T[i]);
        memcpy(b.at<uint32_t>(o_til), tile_off.data(), tile_off.size() * 4);

        tick("tiles");
        /* per-surface records (filled after the lists they point to exist) */
        for (int i = 0; i < n_srf; i++)
        {
            const qr_surface &q = v.srf[i];
            const bool real = is_real(q);
            DSurf d;
            memset(&d, 0, sizeof(d));
            for (int k = 0; k < 3; k++)
            {
                d.pos[k] = q.pos[k]; d.scj[k] = q.scj[k];
                /* an axis without clipping gets an infinite bound: the kernel compares unconditionally */
                d.min[k] = (q.minmax_t & (1u << k)) ? q.min[k] : -__builtin_inff();
                d.max[k] = (q.minmax_t & (1u << (3 + k))) ? q.max[k] : __builtin_inff();
                d.tci[k] = q.tci[k]; d.tcj[k] = q.tcj[k]; d.tck[k] = q.tck[k];
            }
            for (int k = 0; k < 4; k++) d.sci[k] = q.sci[k];
            d.d_eps = q.d_eps; d.t_eps = q.t_eps;
            d.trn = q.trnode != QR_NULL ? b.srf_off(q.trnode) : b.srf_off(n_srf);
            d.props0 = q.props[0]; d.props1 = q.props[1];
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
            DShade h;
            memset(&h, 0, sizeof(h));
            h.srf = b.srf_off(i);
            h.mat[0] = h.mat[1] = b.o_mat + (uint32_t)n_mat * (uint32_t)sizeof(qr_material);
            if (real)
            {
                d.clip = b.compile_clip(i);
                for (int k = 0; k < 2; k++)
                {
                    h.mat[k] = b.o_mat + (uint32_t)q.mat[k] * (uint32_t)sizeof(qr_material);
                    h.lgt[k] = b.compile_lights(q.lst[k * 2]);
                    h.lst[k] = b.compile_list(q.lst[k * 2 + 1]);
                }
            }
            b.at<DSurf>(b.o_srf)[i] = d;
            b.at<DShade>(b.o_shd)[i] = h;
        }
        b.at<DShade>(b.o_shd)[n_srf].srf = b.srf_off(n_srf);
        b.at<DShade>(b.o_shd)[n_srf].mat[0] = b.at<DShade>(b.o_shd)[n_srf].mat[1] = b.o_mat + (uint32_t)n_mat * (uint32_t)sizeof(qr_material);
        for (int k = 0; k < 3; k++) { b.at<DSurf>(b.o_srf)[n_srf].min[k] = -__builtin_inff(); b.at<DSurf>(b.o_srf)[n_srf].max[k] = __builtin_inff(); }
        b.at<DSurf>(b.o_srf)[n_srf].trn = b.srf_off(n_srf);
        tick("surfaces");

        /* wave schedule: one entry {footprint | heaviness << 30, tile-list offset} per wave footprint.
         * heavy = the footprint's tile list holds a reflective or non-opaque surface: those waves can spawn
         * recursion, are started first and get issue priority */
        /* (scratch kept per thread between frames; bound to references once: every use of a thread_local of a shared
         * library is a call) */
        static thread_local std::vector<uint8_t> tl_tile_heavy, tl_empty;
        static thread_local std::vector<int> tl_fb, tl_blk_of;
        static thread_local std::vector<std::vector<uint32_t>> tl_part;    /* [3 * k + class] */
        static thread_local std::vector<uint32_t *> tl_cur;                 /* write cursors into part[] */
        std::vector<uint8_t> &tile_heavy = tl_tile_heavy, &empty = tl_empty;
        std::vector<int> &fb = tl_fb, &blk_of = tl_blk_of;
        std::vector<std::vector<uint32_t>> &part = tl_part;
        std::vector<uint32_t *> &cur = tl_cur;
        tile_heavy.resize(T.size());
        for (size_t t = 0; t < T.size(); t++) tile_heavy[t] = T[t] != QR_NULL ? b.list_heavy[T[t]] : 0;
        /* The drop-in path launches block by block (K horizontal blocks of footprint rows) and copies block k back while block
         * k + 1 renders: the entries are collected per block, recursion-capable footprints first, then the others, then the
         * clear runs */
        const int K = sched_blocks > 1 ? std::min(sched_blocks, std::max(1, nby)) : 1;
        fb.resize((size_t)K + 1); blk_of.resize((size_t)nby);
        for (int k = 0; k <= K; k++) fb[k] = (int)((long long)nby * k / K);
        for (int k = 0; k < K; k++) for (int y = fb[k]; y < fb[k + 1]; y++) blk_of[y] = k;
        if (part.size() < (size_t)K * 3) part.resize((size_t)K * 3);
        cur.resize((size_t)K * 3);
        for (int i = 0; i < K * 3; i++)
        {
            const size_t cap = (size_t)(fb[i / 3 + 1] - fb[i / 3]) * nbx * 2;
            if (part[i].size() < cap) part[i].resize(cap);
            cur[i] = part[i].data();
        }
        tick("s-prep");
        {
            /* footprints are enumerated tile by tile (32x8 pixel groups) to keep neighbours together */
            const int gx = 32 / fw, gy = 8 / fh;
            const bool nest = frm.tile_w == 32 && frm.tile_h == 8;      /* group (tx, ty) IS tile (tx, ty) */
            /* footprints over an empty tile only have zeros to store: runs of them along a row share ONE wave (schedule head
             * = run length, 1..QR_CLEAR_RUN_MAX: values below 256 are no list offsets).  Half of demo scene 1's 32 400
             * footprints at 1080p are such; QR_CLEAR_RUN=1 gives every one its own wave again */
            static const int clear_run = []() { const char *e = getenv("QR_CLEAR_RUN"); const int r = e ? atoi(e) : QR_CLEAR_RUN_MAX;
                                                return r < 1 ? 1 : (r > QR_CLEAR_RUN_MAX ? QR_CLEAR_RUN_MAX : r); }();
            const bool plain = nest && gy == 1;
            if (plain)
            {
                /* the plain frame: a row of tiles is a row of footprints, gx of them per tile with the tile's list */
                for (int by = 0; by < nby; by++)
                {
                    uint32_t **c3 = &cur[(size_t)blk_of[by] * 3];
                    uint32_t *p_hv = c3[0], *p_lt = c3[1], *p_cl = c3[2];
                    const bool in_rows = by < frm.tls_col;
                    const uint32_t *t_off = tile_off.data() + (size_t)by * frm.tls_row;
                    const uint8_t *t_hv = tile_heavy.data() + (size_t)by * frm.tls_row;
                    int run0 = 0, run = 0;                  /* the clear run being collected: first footprint, length */
                    for (int tx = 0; tx * gx < nbx; tx++)
                    {
                        const bool in = in_rows && tx < frm.tls_row;
                        const uint32_t head = in ? t_off[tx] : QR_SCHED_PER_LANE;
                        const uint32_t hv = in ? t_hv[tx] : 0u;
                        const int bx0 = tx * gx, cnt = std::min(gx, nbx - bx0);
                        if (head == 0)
                        {
                            for (int i = 0; i < cnt; i++)
                            {
                                if (run == 0) run0 = bx0 + i;
                                if (++run == clear_run) { p_cl[0] = (uint32_t)run0 | ((uint32_t)by << 14); p_cl[1] = (uint32_t)run; p_cl += 2; run = 0; }
                            }
                            continue;
                        }
                        if (run) { p_cl[0] = (uint32_t)run0 | ((uint32_t)by << 14); p_cl[1] = (uint32_t)run; p_cl += 2; run = 0; }
                        uint32_t *&dst = hv ? p_hv : p_lt;
                        const uint32_t ent = ((uint32_t)by << 14) | ((hv & 3u) << 30);
                        for (int i = 0; i < cnt; i++) { dst[0] = ent | (uint32_t)(bx0 + i); dst[1] = head; dst += 2; }
                    }
                    if (run) { p_cl[0] = (uint32_t)run0 | ((uint32_t)by << 14); p_cl[1] = (uint32_t)run; p_cl += 2; }
                    c3[0] = p_hv; c3[1] = p_lt; c3[2] = p_cl;
                }
            }
            else empty.assign((size_t)nbx * nby, 0);
            if (!plain)
            for (int ty = 0; ty * gy < nby; ty++)
                for (int tx = 0; tx * gx < nbx; tx++)
                {
                    int g_hv = 0; uint32_t g_head = QR_SCHED_PER_LANE;
                    if (nest && tx < frm.tls_row && ty < frm.tls_col)
                    {
                        const size_t t = (size_t)ty * frm.tls_row + tx;
                        g_hv = tile_heavy[t]; g_head = tile_off[t];
                    }
                    for (int j = 0; j < gy; j++)
                    {
                        const int by = ty * gy + j;
                        if (by >= nby) break;
                        uint32_t **row3 = &cur[(size_t)blk_of[by] * 3];
                        if (nest && g_head == 0)
                        {
                            for (int i = 0; i < gx && tx * gx + i < nbx; i++) empty[(size_t)by * nbx + tx * gx + i] = 1;
                            continue;
                        }
                        for (int i = 0; i < gx; i++)
                        {
                            const int bx = tx * gx + i;
                            if (bx >= nbx) break;
                            int hv = g_hv; uint32_t head = g_head;
                            if (!nest)
                            {
                                const int x0 = bx * fw, y0 = by * fh;
                                const int x1 = std::min(x0 + fw - 1, frm.frm_w - 1), y1 = std::min(y0 + fh - 1, frm.frm_h - 1);
                                const int tlx = x0 / frm.tile_w, tly = y0 / frm.tile_h;
                                hv = tile_heavy[(size_t)tly * frm.tls_row + tlx];
                                head = QR_SCHED_PER_LANE;
                                if (tlx == x1 / frm.tile_w && tly == y1 / frm.tile_h) head = tile_off[(size_t)tly * frm.tls_row + tlx];
                                else for (int yy = tly; yy <= y1 / frm.tile_h; yy++) for (int xx = tlx; xx <= x1 / frm.tile_w; xx++) hv |= tile_heavy[(size_t)yy * frm.tls_row + xx];
                            }
                            if (head == 0) { empty[(size_t)by * nbx + bx] = 1; continue; }
                            const uint32_t ent = (uint32_t)bx | ((uint32_t)by << 14) | ((uint32_t)(hv & 3) << 30);
                            uint32_t *&dst = row3[hv ? 0 : 1];
                            dst[0] = ent; dst[1] = head; dst += 2;
                        }
                    }
                }
            tick("s-enum");
            for (int by = 0; by < nby && !plain; by++)
            {
                uint32_t *&dst = cur[(size_t)blk_of[by] * 3 + 2];
                const uint8_t *em = empty.data() + (size_t)by * nbx;
                for (int bx = 0; bx < nbx; )
                {
                    if (!em[bx]) { bx++; continue; }
                    int run = 1;
                    while (run < clear_run && bx + run < nbx && em[bx + run]) run++;
                    dst[0] = (uint32_t)bx | ((uint32_t)by << 14); dst[1] = (uint32_t)run; dst += 2;
                    bx += run;
                }
            }
        }
        tick("s-clear");
        {
            size_t total = 0;
            for (int i = 0; i < K * 3; i++) total += (size_t)(cur[i] - part[i].data());
            if (total > n_sched * 2 || (total & 1)) throw Fail{QR_ERR_ARG, "schedule size mismatch"};
            out.order.resize(total);
            out.block_first.clear(); out.block_row.clear();
            size_t pos = 0;
            for (int k = 0; k < K; k++)
            {
                if (sched_blocks > 1)
                {
                    out.block_first.push_back((uint32_t)(pos / 2));
                    out.block_row.push_back((uint32_t)std::min(fb[k] * fh, frm.frm_h));
                }
                for (int c = 0; c < 3; c++)
                {
                    const size_t cnt = (size_t)(cur[(size_t)k * 3 + c] - part[(size_t)k * 3 + c].data());
                    if (cnt) memcpy(out.order.data() + pos, part[(size_t)k * 3 + c].data(), cnt * 4);
                    pos += cnt;
                }
            }
            if (sched_blocks > 1)
            {
                out.block_first.push_back((uint32_t)(pos / 2));
                out.block_row.push_back((uint32_t)frm.frm_h);
            }
        }
        tick("schedule");
        const size_t n_waves = out.order.size() / 2;        /* schedule entries: one per wave (fewer than footprints: clear runs) */
        memcpy(b.at<uint32_t>(o_ord), out.order.data(), out.order.size() * 4);

        DevHeader &h = *b.at<DevHeader>(0);
        h.fr = frm;
        h.off_shade = b.o_shd; h.off_tiles = o_til; h.off_order = o_ord; h.n_blocks = (uint32_t)n_waves;
        b.alloc(64, 64);                    /* tail padding: wide scalar loads of the last record stay inside */
        out.off_order = o_ord; out.n_sched = (uint32_t)n_waves;
        out.off_srf = b.o_srf; out.off_shade = b.o_shd; out.off_mat = b.o_mat; out.off_lgt = b.o_lgt; out.off_tex = b.o_tex; out.off_tiles = o_til;
        out.n_srf = (uint32_t)n_srf; out.n_mat = (uint32_t)n_mat; out.n_lgt = (uint32_t)n_lgt; out.n_tex = (uint32_t)n_tex; out.n_tiles = (uint32_t)T.size();
        out.off_lists = o_ord + (uint32_t)(n_sched * 8 + 16);
        b.st.bytes = out.blob.size(); b.st.n_grids = b.n_grids; b.st.n_grid_lists = b.n_grid_lists; b.st.n_dda = b.n_dda;
        out.stats = b.st;
        out.has_long_lists = b.any_long;
        out.has_grids = b.n_grids != 0;
        /* box cells and a list the per-lane walkers take (possible only with lowered QR_DDA / QR_GRID thresholds: by default such a
         * list has QR_LONG_CELLS elements and has ended the first attempt already): build again without box cells */
        if (b.box_ok && (b.any_long || b.n_grids != 0)) throw Builder::Restart{};
        b.at<DevHeader>(0)->img_flags = b.any_box ? QR_IMG_BOXES : 0u;
        b.at<DevHeader>(0)->img_bytes = (uint32_t)out.blob.size();
    }
    catch (const Fail &f) { err = f.msg; return f.rc; }
    catch (const Builder::Restart &) { restart = true; return QR_OK; }
    tick("finish");
    if (!verify) return QR_OK;
    const int vrc = qr_program_verify(out, err);
    tick("verify");
    return vrc;
}

int qr_program_build(const qr_scene_view &v, const std::vector<qr_elem> &E, const std::vector<int32_t> &T,
                     const qr_frame &frm, const std::vector<BSphere> &bs, QrProgram &out, std::string &err, int sched_blocks, bool verify)
{
    bool restart = false;
    int rc = program_build_once(v, E, T, frm, bs, out, err, sched_blocks, verify, true, restart);
    if (restart) rc = program_build_once(v, E, T, frm, bs, out, err, sched_blocks, verify, false, restart);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------- */
/* verification of the finished image: every offset the kernel follows                                      */
/* ------------------------------------------------------------------------------------------------------- */

int qr_program_verify(const QrProgram &p, std::string &err)
{
    const std::vector<uint8_t> &b = p.blob;
    const size_t N = b.size();
    auto bad = [&](const char *m) { err = std::string("compiled scene fails verification: ") + m; return QR_ERR_ARG; };
    if (N < sizeof(DevHeader) + 64) return bad("too small");
    const size_t limit = N - 64;        /* records end before the tail padding */
    auto in_arr = [&](uint32_t off, uint32_t base, uint32_t n, size_t sz) {
        return off >= base && (off - base) % sz == 0 && (off - base) / sz <= n && (size_t)off + sz <= limit;
    };
    /* per 32-byte slot of the image: 1 = a list starts here (already checked), 2 = a cell starts here */
    static thread_local std::vector<uint8_t> slot;
    slot.assign(N / 32 + 1, 0);
    /* a list program: cells inside the list area, END-terminated, array ends on cell boundaries inside the run */
    auto check_list = [&](uint32_t off) -> const char * {
        if (off == 0) return nullptr;
        if (off & 8u) return "list offset carries unknown flag bits";
        const bool world = (off & QR_LISTF_WORLD) != 0, dda = (off & QR_LISTF_DDA) != 0, div_ok = (off & QR_LISTF_DIV) != 0;
        off &= ~31u;
        if (off < p.off_lists || (size_t)off + 32 > limit) return "list offset out of range";
        if (slot[off / 32] & 1) return (((slot[off / 32] & 4) != 0) == world && ((slot[off / 32] & 8) != 0) == dda) ? nullptr : "list referenced with different flags";
        slot[off / 32] |= (world ? 5 : 1) | (dda ? 8 : 0);
        uint32_t o = off, end_cell = 0;
        for (;;)
        {
            if ((size_t)o + 32 > limit) return "list runs off the image";
            const CCell *c = (const CCell *)(b.data() + o);
            slot[o / 32] |= 2;
            if (c->op == 0) { end_cell = o; break; }
            o += (c->op & QR_OPT_BV) ? 64 : 32;
        }
        if (world && div_ok)
        {
            /* the length word in front of the program: what walk_pool cuts ranges by */
            uint32_t lw; memcpy(&lw, b.data() + off - 4u, 4);
            bool any_bv = false;
            for (uint32_t q = off; q < end_cell; ) { const CCell *cq = (const CCell *)(b.data() + q); const bool bvq = (cq->op & QR_OPT_BV) != 0; any_bv = any_bv || bvq; q += bvq ? 64 : 32; }
            if ((lw & ~31u) != end_cell - off || (lw & 30u) != 0 || ((lw & 1u) != 0) != !any_bv) return "list length word";
        }
        for (o = off; o < end_cell; )
        {
            const CCell *c = (const CCell *)(b.data() + o);
            const uint32_t t = c->op & QR_OPT_MASK;
            if (t == 0 || (t & (t - 1)) != 0) return "bad opcode";
            if (!in_arr(c->srf, p.off_srf, p.n_srf, sizeof(DSurf)) || c->srf == p.off_srf + p.n_srf * (uint32_t)sizeof(DSurf)) return "cell surface offset out of range";
            if (t == QR_OPT_BV && (c->end <= o + 32 || c->end > end_cell || (c->end & 31) || !(slot[c->end / 32] & 2)))
                return "array end outside its list";
            if (t == QR_OPT_BV)
            {
                const uint32_t mid = ((const CBvExt *)(b.data() + o + 32))->mid;
                if (mid != 0 && (mid <= o + 64 || mid >= c->end || (mid & 31) || !(slot[mid / 32] & 2))) return "array hand-over boundary outside the array";
            }
            if ((c->op & QR_OPF_CACHED) && (c->op & QR_OPF_OWN)) return "bad transform mode";
            if (world && (t == QR_OPT_TRNODE || (c->op & QR_OPF_LOCAL))) return "transform in a list flagged world-space";
            if ((c->op & QR_OPF_CULL) && !(t & (QR_OPT_SOLVER | QR_OPT_BV))) return "cull flag on a cell without solver or volume";
            if (c->op & QR_OPF_BOX)
            {
                if (!(c->op & QR_OPF_CULL) || !(t & QR_OPT_SOLVER) || dda) return "box flag on the wrong kind of cell";
                if (!(c->r2 <= c->cy) || !(c->r2x <= c->cz) || !(c->cx <= c->r) || !(c->r2 > -1e30f) || !(c->r < 1e30f)) return "box cull cell: bad box";
            }
            if ((c->op & QR_OPF_SPHBV) && (t != QR_OPT_BV || !(c->op & QR_OPF_CULL) || (c->op & QR_OPF_LOCAL))) return "sphere-volume flag on the wrong kind of cell";
            if ((c->op & QR_OPF_KX) && (c->op & QR_OPF_KY)) return "bad axis k";
            if ((c->op & QR_OPF_IX) && (c->op & QR_OPF_IY)) return "bad axis i";
            o += (t == QR_OPT_BV) ? 64 : 32;
        }
        if (dda)
        {
            /* the uniform grid in front of the program: tables inside the image, every ref a solver cell of THIS list */
            if (!world || off < p.off_lists + sizeof(CDda)) return "grid record in front of a list that cannot have one";
            const CDda *g = (const CDda *)(b.data() + off - sizeof(CDda));
            const uint32_t nx = g->dims & 255u, ny = (g->dims >> 8) & 255u, nz = (g->dims >> 16) & 255u;
            if ((g->dims >> 24) || nx < 1 || ny < 1 || nz < 1 || nx > QR_GRID_MAX || ny > QR_GRID_MAX || nz > QR_GRID_MAX) return "grid dimensions";
            const size_t nc = (size_t)nx * ny * nz;
            if ((g->cells & 3) || g->cells < p.off_lists || (size_t)g->cells + (nc + 1) * 4 > limit) return "grid cell table";
            if ((g->refs & 31) || g->refs < p.off_lists || (size_t)g->refs + ((size_t)g->n_refs + 2) * sizeof(CCell) > limit) return "grid refs";
            const uint32_t *cs = (const uint32_t *)(b.data() + g->cells);
            if (cs[0] != g->n_out || cs[nc] != g->n_refs) return "grid cell table ends";
            for (size_t q = 0; q < nc; q++) if (cs[q] > cs[q + 1]) return "grid cell table not monotone";
            for (int k = 0; k < 3; k++) if (!(g->size[k] > 0.0f) || !(g->inv[k] > 0.0f) || !(g->size[k] < 1e30f) || !(g->inv[k] < 1e30f)) return "grid cell size";
            const CCell *rf = (const CCell *)(b.data() + g->refs);
            for (uint32_t q = 0; q < g->n_refs; q++)
            {
                const uint32_t t = rf[q].op & QR_OPT_MASK;
                if (t == 0 || (t & (t - 1)) != 0 || !(t & QR_OPT_SOLVER) || (rf[q].op & (QR_OPF_LOCAL | QR_OPF_CLIP))) return "grid ref opcode";
                if (!in_arr(rf[q].srf, p.off_srf, p.n_srf, sizeof(DSurf)) || rf[q].srf == p.off_srf + p.n_srf * (uint32_t)sizeof(DSurf)) return "grid ref surface";
                uint32_t pos; memcpy(&pos, &rf[q].r2x, 4);
                if (pos < off || pos >= end_cell || (pos & 31) || !(slot[pos / 32] & 2)) return "grid ref position";
                const CCell *oc = (const CCell *)(b.data() + pos);
                if (oc->op != rf[q].op || oc->srf != rf[q].srf) return "grid ref is not a copy of its cell";
            }
        }
        return nullptr;
    };
    static const bool vt = getenv("QR_VERIFY_TIMING") != nullptr;
    struct timespec vt0; if (vt) clock_gettime(CLOCK_MONOTONIC, &vt0);
    auto vtick = [&](const char *what) { if (!vt) return; struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "  verify %-8s %.3f ms\n", what, (t1.tv_sec - vt0.tv_sec) * 1e3 + (t1.tv_nsec - vt0.tv_nsec) * 1e-6); vt0 = t1; };
    const DevHeader *h = (const DevHeader *)b.data();
    if (h->off_tiles != p.off_tiles || h->off_order != p.off_order || h->off_shade != p.off_shade) return bad("header offsets");
    if ((size_t)p.off_tiles + (size_t)p.n_tiles * 4 > limit) return bad("tile array");
    const uint32_t *tl = (const uint32_t *)(b.data() + p.off_tiles);
    /* neighbouring tiles mostly share one program (lists are stored once per content): a head equal to the one just checked
     * needs no second look */
    { uint32_t prev = 0; for (uint32_t i = 0; i < p.n_tiles; i++) { if (tl[i] == prev) continue; if (const char *m = check_list(tl[i])) return bad(m); prev = tl[i]; } }
    vtick("tiles");
    if ((size_t)p.off_order + (size_t)p.n_sched * 8 > limit) return bad("schedule");
    const uint32_t *ord = (const uint32_t *)(b.data() + p.off_order);
    const int fw = p.frm.fsaa == 2 ? 4 : 8, fh = p.frm.fsaa == 0 ? 8 : 4;
    const uint32_t max_bx = (uint32_t)((p.frm.frm_w + fw - 1) / fw), max_by = (uint32_t)((p.frm.frm_h + fh - 1) / fh);
    uint32_t prev_hd = 0;
    for (uint32_t i = 0; i < p.n_sched; i++)
    {
        const uint32_t e = ord[2 * i], hd = ord[2 * i + 1];
        if ((e & 0x3FFFu) >= max_bx || ((e >> 14) & 0x3FFFu) >= max_by) return bad("schedule footprint outside the frame");
        if (hd == prev_hd && hd >= 256u) continue;
        if (hd < 256u)
        {
            /* a run of hd empty footprints along the row (0: one) */
            if (hd > QR_CLEAR_RUN_MAX || ((int)(e & 0x3FFFu) + (int)(hd ? hd : 1u) - 1) * fw >= p.frm.frm_w) return bad("clear run leaves the frame");
        }
        else if (hd != QR_SCHED_PER_LANE) { if (const char *m = check_list(hd)) return bad(m); prev_hd = hd; }
    }
    vtick("sched");
    for (uint32_t i = 0; i <= p.n_srf; i++)
    {
        const DSurf *d = (const DSurf *)(b.data() + p.off_srf) + i;
        const DShade *s = (const DShade *)(b.data() + p.off_shade) + i;
        if (!in_arr(d->trn, p.off_srf, p.n_srf, sizeof(DSurf))) return bad("trnode offset");
        if (d->clip != 0)
        {
            if ((d->clip & 31) || d->clip < p.off_lists) return bad("clipper program offset");
            bool in_group = false;
            for (uint32_t o = d->clip;; o += (uint32_t)sizeof(CClip))
            {
                if ((size_t)o + sizeof(CClip) > limit) return bad("clipper program runs off the image");
                const CClip *c = (const CClip *)(b.data() + o);
                if (c->op == 0) { if (in_group) return bad("clipper program ends inside a trnode group"); break; }
                const uint32_t t = c->op & QR_CLT_MASK;
                if (t == 0 || (t & (t - 1)) != 0) return bad("bad clipper opcode");
                if (t != QR_CLT_ENTER && t != QR_CLT_LEAVE && (!in_arr(c->srf, p.off_srf, p.n_srf, sizeof(DSurf)) || c->srf == p.off_srf + p.n_srf * (uint32_t)sizeof(DSurf))) return bad("clipper surface offset");
                if ((c->op & QR_CLF_KX) && (c->op & QR_CLF_KY)) return bad("bad clipper axis");
                if ((c->op & QR_CLF_CACHED) && (c->op & QR_CLF_OWN)) return bad("bad clipper transform mode");
                /* the cached trnode space: opened by a trnode cell, closed behind the cell flagged QR_CLF_LASTC; cached cells
                 * only inside, fast plane cells with one axis mask and a consistent sign */
                if (t == QR_CLT_TRNODE || t == QR_CLT_TRSAME) in_group = true;
                else if ((c->op & QR_CLF_CACHED) && !in_group) return bad("cached clipper cell outside a trnode group");
                else if (!(c->op & QR_CLF_CACHED) && in_group && t != QR_CLT_ENTER && t != QR_CLT_LEAVE) return bad("world-space clipper cell inside a trnode group");
                if (c->op & QR_CLF_LASTC) { if (!in_group) return bad("group end flag outside a group"); in_group = false; }
                if (c->op & QR_CLF_FASTPL)
                {
                    const uint32_t ones = (c->mx == 0xFFFFFFFFu) + (c->my == 0xFFFFFFFFu) + (c->mz == 0xFFFFFFFFu);
                    const uint32_t zeros = (c->mx == 0u) + (c->my == 0u) + (c->mz == 0u);
                    if (t != QR_CLT_PLANE || (c->op & QR_CLF_OWN) || ones != 1 || zeros != 2) return bad("fast plane cell");
                    if (c->sgn != ((c->op & QR_CLF_SGNK) ? 0x80000000u : 0u)) return bad("fast plane cell sign");
                }
            }
        }
        for (int k = 0; k < 2; k++)
        {
            if (!in_arr(s->mat[k], p.off_mat, p.n_mat, sizeof(qr_material))) return bad("material offset");
            if (const char *m = check_list(s->lst[k])) return bad(m);
            if (s->lgt[k] != 0)
            {
                if ((s->lgt[k] & 7) || s->lgt[k] < p.off_lists) return bad("light list offset");
                for (uint32_t o = s->lgt[k];; o += 8)
                {
                    if ((size_t)o + 8 > limit) return bad("light list runs off the image");
                    const CLight *l = (const CLight *)(b.data() + o);
                    const uint32_t lo = l->lgt & ~QR_CLIGHT_LAST;
                    if (!in_arr(lo, p.off_lgt, p.n_lgt, sizeof(qr_light)) || lo == p.off_lgt + p.n_lgt * (uint32_t)sizeof(qr_light)) return bad("light offset");
                    if (l->shadow & QR_LISTF_GRID)
                    {
                        /* shadow lists by hit position: record, table and every list in it */
                        const uint32_t go = l->shadow & ~31u;
                        if ((l->shadow & 31u) != QR_LISTF_GRID || go < p.off_lists || (size_t)go + sizeof(CGrid) > limit) return bad("shadow grid offset");
                        const CGrid *g = (const CGrid *)(b.data() + go);
                        if (g->nx < 1 || g->nx > QR_GRID_MAX || g->ny < 1 || g->ny > QR_GRID_MAX) return bad("shadow grid size");
                        const uint32_t ca = g->comps & 3u, cb = (g->comps >> 2) & 3u;
                        if ((g->comps >> 4) || ca > 2 || cb > 2 || ca == cb) return bad("shadow grid components");
                        if ((g->table & 3) || g->table < p.off_lists || (size_t)g->table + (size_t)g->nx * g->ny * 4 > limit) return bad("shadow grid table");
                        const uint32_t *tb = (const uint32_t *)(b.data() + g->table);
                        for (uint32_t q = 0; q < g->nx * g->ny; q++) if (const char *m = check_list(tb[q])) return bad(m);
                    }
                    else if (const char *m = check_list(l->shadow)) return bad(m);
                    if (l->lgt & QR_CLIGHT_LAST) break;
                }
            }
        }
        if (s->srf != p.off_srf + i * (uint32_t)sizeof(DSurf)) return bad("shade record surface offset");
    }
    vtick("surfaces");
    for (uint32_t i = 0; i <= p.n_mat; i++)
    {
        const qr_material *m = (const qr_material *)(b.data() + p.off_mat) + i;
        const uint64_t n = (uint64_t)(m->xmask + 1) * (m->ymask + 1);
        if ((uint32_t)m->tex < p.off_tex || ((uint32_t)m->tex & 3) || (uint64_t)(uint32_t)m->tex + n * 4 > (uint64_t)p.off_tex + ((uint64_t)p.n_tex + 1) * 4)
            return bad("texture offset");
    }
    return QR_OK;
}

/* host-only entry point: validate + compile a snapshot without a device (used by the CPU test-suite) */
extern "C" int qr_program_stats(const void *blob, uint64_t size, qr_program_info *info)
{
    if (blob == nullptr || info == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    qr_scene_view v;
    const int rc0 = qr_scene_view_init(&v, blob, size);
    if (rc0 != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot (qr_scene_view_init " + std::to_string(rc0) + ")");
    std::string err;
    static const bool timing = getenv("QR_COMPILE_TIMING") != nullptr;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto tick = [&](const char *what) {
        if (!timing) return;
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "  compile %-10s %.3f ms\n", what, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
        ts0 = t1;
    };
    int rc = qr_snapshot_validate(v, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    tick("validate");
    std::vector<BSphere> bs;
    qr_bound_spheres(v, bs);
    tick("bounds");
    std::vector<qr_elem> E(v.elm, v.elm + v.hdr->n_elm);
    std::vector<int32_t> T(v.tiles, v.tiles + v.hdr->n_tiles);
    tick("copies");
    QrProgram p;
    rc = qr_program_build(v, E, T, *v.frame, bs, p, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    if (const char *dump = getenv("QR_DUMP_IMAGE"))        /* debugging aid: the device image as a file */
        if (FILE *f = fopen(dump, "wb")) { fwrite(p.blob.data(), 1, p.blob.size(), f); fclose(f); }
    memset(info, 0, sizeof(*info));
    info->bytes = p.stats.bytes; info->n_lists = p.stats.n_lists; info->n_cells = p.stats.n_cells;
    info->n_dropped = p.stats.n_dropped; info->n_clip_cells = p.stats.n_clip_cells; info->n_sched = p.n_sched;
    info->n_grids = p.stats.n_grids; info->n_grid_lists = p.stats.n_grid_lists; info->n_dda = p.stats.n_dda;
    return QR_OK;
}

/* ------------------------------------------------------------------------------------------------------- */
/* per-surface list building (the role of rt_SceneThread::ssort / lsort, engine.cpp:2134-2753)             */
/* ------------------------------------------------------------------------------------------------------- */

/*
 * For scenes that arrive with one global surface list only (scenes the reference engine did not prepare: the
 * synthetic 10k-quadric scene, user scenes), build what the engine's ssort / lsort build with their bounding-box
 * predicates (bbox_shad rtgeom.cpp:1004, bbox_side rtgeom.cpp:1954):
 *   - per surface and light a SHADOW list: the members of the global list that can stand between the light
 *     and the surface -- here: whose conservative bounding sphere (qr_bounds.hpp) meets the hull of the light's
 *     position and the surface's bounding sphere;
 *   - per surface side a REFLECTION / REFRACTION list: for an untransformed plane the members that reach into
 *     the half-space that side faces; everything for other shapes (what the engine does for them too);
 *   - light lists: every light, with its shadow list.
 * Lists only cull (a shorter list must not change any pixel), so the predicates are our own conservative ones, not
 * the engine's box arithmetic.  The global list's ORDER and STRUCTURE are kept: members keep their relative order,
 * a bounding-volume or trnode element stays in front of its first kept member with `data` patched to the last
 * kept one -- the format the walk (and the reference) expects.  Arrays whose union sphere fails the predicate are
 * pruned without visiting their members.  Output: a new snapshot.
 */
#ifndef QR_FLAT_LIST_MAX
#define QR_FLAT_LIST_MAX 128
#endif

/*
 * Per-side placement (qr_sides.cpp = the engine's bbox_side / clip_side): NOT a cull in the engine but part of its picture.
 * Lights: with RT_OPTS_2SIDED its lsort (engine.cpp:2503-2533) enters a light on the outer and / or the inner light list by
 * where it stands relative to the clipped surface, so the inside of a closed convex shell never gets a light that stands
 * outside, whether or not the shell casts a shadow.  Surfaces: ssort (engine.cpp:2134-2450) puts every node on the outer
 * and / or inner list of a surface whose material reflects or is not opaque; a convex shape is absent from its own outer
 * list, a bowl's inside holds what the box test sees through its opening.  The engine's per-side lists are copies of its
 * hierarchy per surface -- memory grows with the square of the scene (engine.cpp:2951-2956) -- so scenes with more than
 * QR_SIDES_EXACT_MAX real surfaces keep ONE shared list for both sides of their quadrics (a superset of what the engine would
 * place there; planes get the half-space filter below).
 */
#ifndef QR_SIDES_EXACT_MAX
#define QR_SIDES_EXACT_MAX 512
#endif

int qr_snapshot_build_lists(const qr_scene_view &v, std::vector<uint8_t> &out, std::string &err)
{
    int rc = qr_snapshot_validate(v, err);
    if (rc != QR_OK) return rc;
    const qr_frame &fr = *v.frame;
    if (fr.clist == QR_NULL) { err = "snapshot has no global list (clist)"; return QR_ERR_ARG; }
    std::vector<BSphere> bs;
    qr_bound_spheres(v, bs);
    const int n_srf = (int)v.hdr->n_srf, n_lgt = (int)v.hdr->n_lgt;
    std::vector<qr_elem> E(v.elm, v.elm + v.hdr->n_elm);
    std::vector<qr_surface> S(v.srf, v.srf + n_srf);
    ListFilter lf(v, bs, E, fr.clist);
    QrSideGeom geom(v);
    int exact_max = QR_SIDES_EXACT_MAX;
    if (const char *e = getenv("QR_SIDES_EXACT_MAX")) exact_max = atoi(e);
    const bool exact_sides = lf.n_real <= exact_max;
    std::vector<uint8_t> mask(lf.ch.size());
    /* boxes of the chain's array elements (transform nodes; bounding-volume nodes where the list has them: the engine drops
     * them from a camera list when its screen tiling is on, engine.cpp:1711-1725) */
    std::vector<int> head_box(lf.ch.size(), -1);
    if (exact_sides)
    {
        std::vector<int> leaves;
        for (size_t k = 0; k < lf.ch.size(); k++)
        {
            if (!lf.ch[k].head) continue;
            leaves.clear();
            for (int q = (int)k + 1; q <= lf.ch[k].last; q++)
                if (!lf.ch[q].head && is_real(v.srf[lf.ch[q].si])) leaves.push_back(lf.ch[q].si);
            const qr_surface &rec = v.srf[lf.ch[k].si];
            int space = rec.srf_t[3] < 0 ? lf.ch[k].si : rec.trnode;          /* a transform node's box lives in its own space */
            if (rec.srf_t[3] >= 0 && rec.trnode == QR_NULL && rec.sci[3] != 0.75f && !leaves.empty())
            {
                /* sphere form of a transform node's inner box (object.cpp:2268-2289): the box is in the members' space */
                int t = v.srf[leaves[0]].trnode;
                for (int m : leaves) if (v.srf[m].trnode != t) t = QR_NULL;
                space = t;
            }
            head_box[k] = geom.add_array_box(leaves.data(), (int)leaves.size(), space);
        }
    }

    /* one surface: its two side lists, its light lists with their shadow lists.  Appends to the element array it is given */
    auto one_surface = [&](int i, ListFilter &lf, std::vector<qr_elem> &E, std::vector<uint8_t> &mask, std::vector<qr_surface> &S)
    {
        qr_surface &s = S[i];
        if (!is_real(s)) return;
        const BSphere &sb = bs[i];
        const bool s_bounded = sb.r < 1e18f;
        /* reflection / refraction lists per side */
        int side_list[2] = { fr.clist, fr.clist };
        if (exact_sides)
        {
            /* the walk of ssort / lsort over the hierarchy, engine.cpp:2222-2330, 2575-2680: a node's sides come from
             * bbox_side -- unless an array above it was seen from one side only: then everything under that array goes where
             * the array went, unasked; an array seen from no side (every corner within the margin of a plane) is not entered
             * at all.  The same placement serves the surface lists and, below, the shadow list of each side's light entries */
            int inherit_until = -1; uint8_t c = 3;
            for (int k = 0; k < (int)lf.ch.size(); k++)
            {
                if (k > inherit_until)
                {
                    inherit_until = -1;
                    const ListFilter::Node &nd = lf.ch[k];
                    if (nd.head) c = (uint8_t)geom.side(head_box[k], i);
                    else c = is_real(v.srf[nd.si]) ? (uint8_t)geom.side(nd.si, i) : (uint8_t)3;
                    if (nd.head && c < 3) inherit_until = nd.last;
                }
                mask[k] = c;
            }
            /* surfaces whose material neither reflects nor lets light through keep the global list on both sides (no ray
             * ever walks it, engine.cpp:2155-2176) */
            if (geom.builds_side_lists(i))
                for (int side = 0; side < 2; side++)
                {
                    const uint8_t bit = side == 0 ? 2 : 1;          /* side 0 = outer */
                    side_list[side] = lf.filter_nodes([&](int k) { return (mask[k] & bit) != 0; });
                }
        }
        else if (s.srf_t[0] == 1 && s.has_trm == 0)
        {
            const int k = (int)((s.axes >> 4) & 3);
            const double sg = ((s.axes >> 10) & 1) ? -1.0 : 1.0;
            for (int side = 0; side < 2; side++)
            {
                /* side 0 (outer) is hit by rays travelling against the signed axis: it faces +sg along k */
                const double face = side == 0 ? sg : -sg;
                side_list[side] = lf.filter([&](const BSphere &x, bool bounded) {
                    if (!bounded) return true;
                    return ((double)x.c[k] - (double)s.pos[k]) * face + (double)x.r * 1.001 + 1e-3 >= 0.0;
                });
            }
        }
        s.lst[1] = side_list[0]; s.lst[3] = side_list[1];
        /* light lists of the two sides; which side a light is entered on is the engine's rule (clip_side), not a cull */
        int l_head[2] = { QR_NULL, QR_NULL }, l_tail[2] = { QR_NULL, QR_NULL };
        for (int l = 0; l < n_lgt; l++)
        {
            /* what may stand between the light and the surface (bbox_shad's role): the hull of the light's position and the
             * surface's bounding sphere; an unbounded surface, or a light inside / next to that sphere, keeps everything */
            const double sc[3] = { sb.c[0], sb.c[1], sb.c[2] };
            const HullPred hp(v.lgt[l].pos, sc, s_bounded ? (double)sb.r : 1e30);
            int shadow = fr.clist;
            if (!exact_sides && !hp.all) shadow = lf.filter(hp);
            const int sides = geom.clip_side(i, v.lgt[l].pos);
            for (int side = 0; side < 2; side++)
            {
                const uint8_t bit = side == 0 ? 2 : 1;
                if (!(sides & bit)) continue;
                /* the engine keeps one shadow list per side a light is entered on: what may cast a shadow (its bbox_shad; here
                 * the hull predicate above) AND is placed on that side of the surface (lsort, engine.cpp:2590-2612) -- the
                 * outer side of a convex shape never tests the shape itself */
                int side_shadow = shadow;
                if (exact_sides)
                    side_shadow = lf.filter_nodes([&](int k) {
                        if (!(mask[k] & bit)) return false;
                        const ListFilter::Node &nd = lf.ch[k];
                        if (!nd.head && !is_real(v.srf[nd.si])) return true;
                        /* the engine's own box predicate, not a tighter or looser one of ours: where it misses a caster (its
                         * corner / face / edge tests are not exhaustive) the reference's picture has no shadow there */
                        return geom.shad(v.lgt[l].pos, nd.head ? head_box[k] : nd.si, i) != 0;
                    });
                qr_elem c; c.simd = l; c.data = side_shadow; c.next = QR_NULL; c.kind = 0;
                E.push_back(c);
                const int ix = (int)E.size() - 1;
                if (l_tail[side] != QR_NULL) E[l_tail[side]].next = ix; else l_head[side] = ix;
                l_tail[side] = ix;
            }
        }
        s.lst[0] = l_head[0]; s.lst[2] = l_head[1];
    };

    /*
     * Large scenes (the hull-predicate path): surfaces are independent of each other, so ranges of them go to worker threads,
     * each with its own copy of the source elements to append to; the appended runs are then concatenated in surface order and
     * their indices shifted -- element for element the array a single thread builds (10 000 objects x 4 lights: 0.67 s on one
     * core).  QR_HOST_THREADS=1: one thread.  The exact path (<= QR_SIDES_EXACT_MAX surfaces) is small and stays serial.
     */
    int n_thr = 1;
    if (!exact_sides)
    {
        const char *te = getenv("QR_HOST_THREADS");
        n_thr = te ? atoi(te) : (int)std::thread::hardware_concurrency();
        if (n_thr > 16) n_thr = 16;
        if (n_thr > n_srf / 64) n_thr = n_srf / 64;
        if (n_thr < 1) n_thr = 1;
    }
    if (n_thr == 1)
    {
        for (int i = 0; i < n_srf; i++) one_surface(i, lf, E, mask, S);
    }
    else
    {
        const size_t n0 = E.size();
        std::vector<std::vector<qr_elem>> part((size_t)n_thr);
        std::vector<std::thread> pool;
        std::vector<int> first((size_t)n_thr + 1);
        for (int t = 0; t <= n_thr; t++) first[(size_t)t] = (int)((long long)n_srf * t / n_thr);
        for (int t = 0; t < n_thr; t++)
            pool.emplace_back([&, t]() {
                std::vector<qr_elem> Et(E.begin(), E.begin() + (ptrdiff_t)n0);
                ListFilter lft(v, bs, Et, fr.clist);
                std::vector<uint8_t> mk(lft.ch.size());
                for (int i = first[(size_t)t]; i < first[(size_t)t + 1]; i++) one_surface(i, lft, Et, mk, S);
                part[(size_t)t].assign(Et.begin() + (ptrdiff_t)n0, Et.end());
            });
        for (std::thread &th : pool) th.join();
        for (int t = 0; t < n_thr; t++)
        {
            /* indices >= n0 of this part are its own appended elements: they move to where the part lands */
            const int shift = (int)E.size() - (int)n0;
            auto fix = [&](int32_t &x) { if (x >= (int32_t)n0) x += shift; };
            for (qr_elem c : part[(size_t)t]) { fix(c.next); fix(c.data); E.push_back(c); }
            for (int i = first[(size_t)t]; i < first[(size_t)t + 1]; i++)
                if (is_real(S[(size_t)i])) for (int k = 0; k < 4; k++) fix(S[(size_t)i].lst[k]);
        }
    }

    /* serialise the new snapshot: same sections, larger element array */
    qr_header h = *v.hdr;
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = a16(sizeof(qr_header));
    h.header_bytes = sizeof(qr_header);
    h.off_frame = (uint32_t)off;  off = a16(off + sizeof(qr_frame));
    h.off_srf = (uint32_t)off;    off = a16(off + (size_t)n_srf * sizeof(qr_surface));
    h.off_mat = (uint32_t)off;    off = a16(off + (size_t)h.n_mat * sizeof(qr_material));
    h.off_lgt = (uint32_t)off;    off = a16(off + (size_t)h.n_lgt * sizeof(qr_light));
    h.off_elm = (uint32_t)off;    off = a16(off + E.size() * sizeof(qr_elem));
    h.off_tiles = (uint32_t)off;  off = a16(off + (size_t)h.n_tiles * 4);
    h.off_texels = (uint32_t)off; off = a16(off + (size_t)h.n_texels * 4);
    if (off > 0xFFFFFFFFull) { err = "snapshot exceeds 4 GiB"; return QR_ERR_NOMEM; }
    h.n_elm = (uint32_t)E.size();
    h.total_bytes = (uint32_t)off;
    out.assign(off, 0);
    memcpy(out.data(), &h, sizeof(h));
    memcpy(out.data() + h.off_frame, v.frame, sizeof(qr_frame));
    memcpy(out.data() + h.off_srf, S.data(), S.size() * sizeof(qr_surface));
    if (h.n_mat) memcpy(out.data() + h.off_mat, v.mat, (size_t)h.n_mat * sizeof(qr_material));
    if (h.n_lgt) memcpy(out.data() + h.off_lgt, v.lgt, (size_t)h.n_lgt * sizeof(qr_light));
    memcpy(out.data() + h.off_elm, E.data(), E.size() * sizeof(qr_elem));
    if (h.n_tiles) memcpy(out.data() + h.off_tiles, v.tiles, (size_t)h.n_tiles * 4);
    if (h.n_texels) memcpy(out.data() + h.off_texels, v.texels, (size_t)h.n_texels * 4);
    return QR_OK;
}

extern "C" int qr_snapshot_build_lists_c(const void *blob, uint64_t size, void **out_blob, uint64_t *out_size)
{
    if (blob == nullptr || out_blob == nullptr || out_size == nullptr) return qr_fail(QR_ERR_ARG, "null argument");
    qr_scene_view v;
    const int rc0 = qr_scene_view_init(&v, blob, size);
    if (rc0 != 0) return qr_fail(QR_ERR_ARG, "malformed snapshot (qr_scene_view_init " + std::to_string(rc0) + ")");
    std::vector<uint8_t> out;
    std::string err;
    const int rc = qr_snapshot_build_lists(v, out, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    void *p = malloc(out.size());
    if (p == nullptr) return qr_fail(QR_ERR_NOMEM, "out of memory");
    memcpy(p, out.data(), out.size());
    *out_blob = p; *out_size = out.size();
    return QR_OK;
}
