/*
 * qr_walker.cpp - flatten the reference's rt_SIMD_INFOX pointer graph.
 *
 * Host-side half of the drop-in boundary: decodes the SIMD-laid-out structures
 * of core/tracer/tracer.h (rt_ELEM 127-141, rt_SIMD_INFOX 150-407,
 * rt_SIMD_CONTEXT 426-662, rt_SIMD_CAMERA 677-755, rt_SIMD_LIGHT 765-811,
 * rt_SIMD_SURFACE 821-969, rt_SIMD_MATERIAL 979-1078) WITHOUT including any
 * reference header: all offsets are recomputed from the DP(Q*0x..+0x..*P)
 * formulas given there, parameterised by qr_abi_desc {Q, P}.
 *
 * Output: one contiguous blob in the layout of include/qr_scene.h.
 *
 * No HIP in this file; it is also linked into the in-container capture driver.
 */
#include "qr_internal.h"

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>

namespace {

struct Abi
{
    uint32_t Q;     /* quads */
    uint32_t P;     /* pointer size in 32-bit words */
    size_t   ps;    /* pointer slot bytes */
};

inline uint64_t rd_ptr(const Abi &a, const uint8_t *base, size_t off)
{
    if (a.P == 2) { uint64_t v; memcpy(&v, base + off, 8); return v; }
    uint32_t v; memcpy(&v, base + off, 4); return v;
}
inline int64_t rd_cell(const Abi &a, const uint8_t *base, size_t off)
{
    if (a.P == 2) { int64_t v; memcpy(&v, base + off, 8); return v; }
    int32_t v; memcpy(&v, base + off, 4); return v;
}
inline int32_t rd_i32(const uint8_t *base, size_t off)
{
    int32_t v; memcpy(&v, base + off, 4); return v;
}
inline uint32_t rd_u32(const uint8_t *base, size_t off)
{
    uint32_t v; memcpy(&v, base + off, 4); return v;
}
inline float rd_f32(const uint8_t *base, size_t off)
{
    float v; memcpy(&v, base + off, 4); return v;
}

/* rt_ELEM, tracer.h:127-141: four P-sized slots */
struct ElemRaw { int64_t data; uint64_t simd; uint64_t next; };
inline ElemRaw rd_elem(const Abi &a, uint64_t p)
{
    const uint8_t *e = (const uint8_t *)(uintptr_t)p;
    ElemRaw r;
    r.data = rd_cell(a, e, 0 * a.ps);
    r.simd = rd_ptr (a, e, 1 * a.ps);
    r.next = rd_ptr (a, e, 3 * a.ps);
    return r;
}

enum ListKind { LIST_SURFACES, LIST_CLIPPERS, LIST_LIGHTS };

/*
 * pointer -> dense index map: open addressing, linear probing, grows by doubling.  The element map
 * sees every list cell of the frame (tens of thousands per call, every frame), where node-based
 * std::unordered_map spent most of the flatten time in its allocator.
 */
class PtrMap
{
    /* Which slots are in use is kept apart, one bit per slot: the flattener mostly looks up cells it has NOT seen (every list
     * cell of a frame is entered once), and such a lookup then ends in the bit set -- a few KB that stay in the first-level
     * cache -- without reading the table; clear() zeroes the bits and leaves the table as it is. */
    struct Slot { uint64_t key; int32_t val; int32_t pad; };
    std::vector<Slot> tab;
    std::vector<uint64_t> occ;
    size_t used = 0;
    static size_t slot(uint64_t k, size_t mask) { return (size_t)((k >> 4) * 0x9E3779B97F4A7C15ull >> 17) & mask; }
    bool in_use(size_t j) const { return (occ[j >> 6] >> (j & 63)) & 1u; }
    void rebuild(size_t n)
    {
        std::vector<Slot> t(n, Slot{0, 0, 0});
        std::vector<uint64_t> o(n / 64, 0);
        for (size_t i = 0; i < tab.size(); i++)
            if (in_use(i))
            {
                size_t j = slot(tab[i].key, n - 1);
                while ((o[j >> 6] >> (j & 63)) & 1u) j = (j + 1) & (n - 1);
                t[j] = tab[i]; o[j >> 6] |= (uint64_t)1 << (j & 63);
            }
        tab.swap(t); occ.swap(o);
    }
public:
    void clear() { if (used) std::fill(occ.begin(), occ.end(), (uint64_t)0); used = 0; }
    const int32_t *find(uint64_t key) const
    {
        if (tab.empty()) return nullptr;
        const size_t mask = tab.size() - 1;
        for (size_t j = slot(key, mask); in_use(j); j = (j + 1) & mask) if (tab[j].key == key) return &tab[j].val;
        return nullptr;
    }
    void reserve(size_t n)
    {
        size_t cap = 1024; while (cap < 2 * n) cap *= 2;
        if (cap > tab.size()) rebuild(cap);
    }
    /* the value stored for key, or -- when there is none -- val after storing it (fresh = true) */
    int32_t find_or_put(uint64_t key, int32_t val, bool &fresh)
    {
        if ((used + 1) * 2 > tab.size()) rebuild(tab.empty() ? 1024 : tab.size() * 2);
        const size_t mask = tab.size() - 1;
        size_t j = slot(key, mask);
        for (; in_use(j); j = (j + 1) & mask) if (tab[j].key == key) { fresh = false; return tab[j].val; }
        used++; tab[j] = Slot{key, val, 0}; occ[j >> 6] |= (uint64_t)1 << (j & 63); fresh = true;
        return val;
    }
    void put(uint64_t key, int32_t val)
    {
        bool fresh;
        if (find_or_put(key, val, fresh) != val)
        {
            const size_t mask = tab.size() - 1;
            for (size_t j = slot(key, mask); ; j = (j + 1) & mask) if (in_use(j) && tab[j].key == key) { tab[j].val = val; break; }
        }
    }
    int32_t at(uint64_t key) const { const int32_t *p = find(key); return p ? *p : QR_NULL; }
    void by_index(std::vector<uint64_t> &out, size_t n) const      /* out[value] = key */
    {
        out.assign(n, 0);
        for (size_t i = 0; i < tab.size(); i++) if (in_use(i) && (size_t)tab[i].val < n) out[(size_t)tab[i].val] = tab[i].key;
    }
};

struct Walker
{
    Abi a;
    bool pt = false;            /* inf_PT_ON: materials carry their emission */
    std::vector<qr_surface>  srf;
    std::vector<qr_material> mat;
    std::vector<qr_light>    lgt;
    std::vector<qr_elem>     elm;
    std::vector<uint32_t>    texels;
    PtrMap srf_ix, mat_ix, lgt_ix, elm_ix, tex_ix;
    std::deque<uint64_t> srf_todo;
    std::vector<uint64_t> fresh_pool;       /* walk_list's new cells, stacked (a light list walks shadow lists inside) */
    std::vector<int32_t> tiles;
    std::string err;

    /* The storage of a thread's walkers goes from frame to frame: a Walker is a local object (the optimiser keeps its fields
     * in registers; one reached through a thread_local reference ran 1.5x slower) that borrows the vectors and maps of the
     * thread's store for the call and hands them back emptied, capacity kept. */
    void swap_storage(Walker &o)
    {
        srf.swap(o.srf); mat.swap(o.mat); lgt.swap(o.lgt); elm.swap(o.elm); texels.swap(o.texels);
        std::swap(srf_ix, o.srf_ix); std::swap(mat_ix, o.mat_ix); std::swap(lgt_ix, o.lgt_ix); std::swap(elm_ix, o.elm_ix); std::swap(tex_ix, o.tex_ix);
        fresh_pool.swap(o.fresh_pool); tiles.swap(o.tiles);
    }
    void reset()
    {
        srf.clear(); mat.clear(); lgt.clear(); elm.clear(); texels.clear();
        srf_ix.clear(); mat_ix.clear(); lgt_ix.clear(); elm_ix.clear(); tex_ix.clear();
        srf_todo.clear(); fresh_pool.clear(); tiles.clear(); err.clear();
        pt = false;
    }

    /* surface tag, srf_SRF_T(TAG) = DP(Q*0x240 + 0x0C), tracer.h:957-958 */
    int32_t srf_tag(uint64_t p) const
    {
        return rd_i32((const uint8_t *)(uintptr_t)p, a.Q * 0x240 + 0x0C);
    }

    int32_t get_srf(uint64_t p)
    {
        if (p == 0) return QR_NULL;
        if (const int32_t *it = srf_ix.find(p)) return *it;
        int32_t ix = (int32_t)srf.size();
        srf_ix.put(p, ix);
        qr_surface s; memset(&s, 0, sizeof(s));
        srf.push_back(s);
        srf_todo.push_back(p);
        return ix;
    }

    int32_t get_lgt(uint64_t p)
    {
        if (p == 0) return QR_NULL;
        if (const int32_t *it = lgt_ix.find(p)) return *it;
        const uint8_t *l = (const uint8_t *)(uintptr_t)p;
        const size_t q = a.Q * 0x10;
        qr_light o; memset(&o, 0, sizeof(o));
        o.t_max  = rd_f32(l, q * 0x0);
        o.pos[0] = rd_f32(l, q * 0x1);
        o.pos[1] = rd_f32(l, q * 0x2);
        o.pos[2] = rd_f32(l, q * 0x3);
        o.col[0] = rd_f32(l, q * 0x4);
        o.col[1] = rd_f32(l, q * 0x5);
        o.col[2] = rd_f32(l, q * 0x6);
        o.l_src  = rd_f32(l, q * 0x7);
        o.a_qdr  = rd_f32(l, q * 0x8);
        o.a_lnr  = rd_f32(l, q * 0x9);
        o.a_cnt  = rd_f32(l, q * 0xA);
        o.a_rng  = rd_f32(l, q * 0xB);
        int32_t ix = (int32_t)lgt.size();
        lgt.push_back(o);
        lgt_ix.put(p, ix);
        return ix;
    }

    int32_t get_tex(uint64_t p, uint32_t xdim, uint32_t ydim)
    {
        if (p == 0) return QR_NULL;
        if (const int32_t *it = tex_ix.find(p)) return *it;
        int32_t off = (int32_t)texels.size();
        const uint32_t *t = (const uint32_t *)(uintptr_t)p;
        size_t n = (size_t)xdim * ydim;
        texels.insert(texels.end(), t, t + n);
        tex_ix.put(p, off);
        return off;
    }

    int32_t get_mat(uint64_t p)
    {
        if (p == 0) return QR_NULL;
        if (const int32_t *it = mat_ix.find(p)) return *it;
        const uint8_t *m = (const uint8_t *)(uintptr_t)p;
        const size_t q = a.Q * 0x10;
        qr_material o; memset(&o, 0, sizeof(o));
        o.xscal = rd_f32(m, q * 0x00);
        o.yscal = rd_f32(m, q * 0x01);
        o.xoffs = rd_f32(m, q * 0x02);
        o.yoffs = rd_f32(m, q * 0x03);
        o.xmask = rd_u32(m, q * 0x04);
        o.ymask = rd_u32(m, q * 0x05);
        o.yshft = rd_u32(m, q * 0x06);             /* yshft[0], object.cpp:4126-4127 */
        uint64_t tex = rd_ptr(a, m, q * 0x07);      /* mat_TEX_P */
        /* t_map holds byte offsets axis*Q*16 relative to ctx_TEX_O (object.cpp:4099-4100) */
        o.t_map[0] = rd_i32(m, q * 0x08 + 0) / (int32_t)q;
        o.t_map[1] = rd_i32(m, q * 0x08 + 4) / (int32_t)q;
        o.l_dff = rd_f32(m, q * 0x0A);
        o.l_spc = rd_f32(m, q * 0x0B);
        o.l_pow = rd_u32(m, q * 0x0C);              /* l_pow[0] */
        o.c_rfl = rd_f32(m, q * 0x0E);
        o.c_trn = rd_f32(m, q * 0x0F);
        o.c_rfr = rd_f32(m, q * 0x10);
        o.rfr_2 = rd_f32(m, q * 0x11);
        o.c_rcp = rd_f32(m, q * 0x12);
        o.ext_2 = rd_f32(m, q * 0x13);
        o.clamp = rd_f32(m, q * 0x14);
        o.cmask = rd_u32(m, q * 0x15);
        if (pt) { o.emis[0] = rd_f32(m, q * 0x17); o.emis[1] = rd_f32(m, q * 0x18); o.emis[2] = rd_f32(m, q * 0x19); }   /* mat_COL_R/G/B */
        if (o.xmask > 0xFFFF || o.ymask > 0xFFFF)
        {
            err = "material texture dimensions out of range";
            return QR_NULL;
        }
        o.tex = get_tex(tex, o.xmask + 1, o.ymask + 1);
        int32_t ix = (int32_t)mat.size();
        mat.push_back(o);
        mat_ix.put(p, ix);
        return ix;
    }

    /*
     * Flatten the list starting at element pointer `head`.
     * Pass 1 assigns dense indices along `next` (stopping at an element that is
     * already known: shared tails / shared whole lists are kept shared),
     * pass 2 fills the records, so that `data` of an array element can refer
     * to a later element of the same list.
     */
    int32_t walk_list(uint64_t head, ListKind kind)
    {
        if (head == 0) return QR_NULL;

        /* cells new to this walk get consecutive indices from `base`, so inside the run `next` is the
         * following index and only the run's exit (NULL or a cell met before) needs a lookup: this runs
         * for every tile list of every frame in the drop-in path */
        const size_t f0 = fresh_pool.size();
        const int32_t base = (int32_t)elm.size();
        int32_t exit_ix = QR_NULL;
        for (uint64_t p = head; p != 0; )
        {
            bool is_new;
            const int32_t ix = elm_ix.find_or_put(p, (int32_t)elm.size(), is_new);
            if (!is_new) { exit_ix = ix; break; }
            qr_elem e; e.simd = QR_NULL; e.data = QR_NULL; e.next = QR_NULL; e.kind = 0;
            elm.push_back(e);
            fresh_pool.push_back(p);
            p = rd_elem(a, p).next;
            if (elm.size() > (size_t)64 * 1024 * 1024) { err = "element list too long / cyclic"; return QR_NULL; }
        }
        const size_t n_fresh = fresh_pool.size() - f0;
        if (n_fresh == 0) return exit_ix;           /* the head is a cell met before: a list (or tail) shared with an earlier walk */

        for (size_t fi = 0; fi < n_fresh; fi++)
        {
            const uint64_t p = fresh_pool[f0 + fi];
            ElemRaw r = rd_elem(a, p);
            qr_elem e;
            e.next = fi + 1 < n_fresh ? base + (int32_t)fi + 1 : exit_ix;
            e.kind = 0;
            e.data = QR_NULL;
            e.simd = QR_NULL;
            switch (kind)
            {
            case LIST_SURFACES:
            {
                /* engine.cpp:1671-1694: surface -> data 0; array -> last|type */
                e.simd = get_srf(r.simd);
                e.kind = (int32_t)(r.data & 3);
                uint64_t last = (uint64_t)r.data & ~(uint64_t)3;
                if (last != 0)
                {
                    const int32_t *it = elm_ix.find(last);
                    if (!it) { err = "array element's last element is outside its list"; return QR_NULL; }
                    e.data = *it;
                }
                break;
            }
            case LIST_CLIPPERS:
            {
                /* engine.cpp:1845-1947 */
                if (r.simd == 0)
                {
                    e.data = (int32_t)r.data;           /* accum marker -1 / +1 */
                }
                else
                {
                    e.simd = get_srf(r.simd);
                    if (srf_tag(r.simd) < 0)
                    {
                        const int32_t *it = elm_ix.find((uint64_t)r.data);
                        if (!it) { err = "clip trnode's last element is outside its list"; return QR_NULL; }
                        e.data = *it;
                        e.kind = 2;                     /* trnode marker in clip lists */
                    }
                    else
                    {
                        e.data = (int32_t)r.data;       /* clip side */
                    }
                }
                break;
            }
            case LIST_LIGHTS:
            {
                /* engine.cpp:1126: data -> shadow list */
                e.simd = get_lgt(r.simd);
                e.data = QR_NULL;                       /* filled below, may recurse */
                break;
            }
            }
            elm[(size_t)base + fi] = e;
        }

        if (kind == LIST_LIGHTS)
        {
            for (size_t fi = 0; fi < n_fresh; fi++)
            {
                ElemRaw r = rd_elem(a, fresh_pool[f0 + fi]);
                int32_t sh = walk_list((uint64_t)r.data, LIST_SURFACES);
                elm[(size_t)base + fi].data = sh;
            }
        }
        fresh_pool.resize(f0);
        return base;
    }

    void flatten_surface(uint64_t p)
    {
        const uint8_t *s = (const uint8_t *)(uintptr_t)p;
        const size_t q = a.Q * 0x10;
        qr_surface o; memset(&o, 0, sizeof(o));

        o.c_def  = rd_u32(s, q * 0x00);
        o.pos[0] = rd_f32(s, q * 0x01); o.pos[1] = rd_f32(s, q * 0x02); o.pos[2] = rd_f32(s, q * 0x03);
        o.min[0] = rd_f32(s, q * 0x04); o.min[1] = rd_f32(s, q * 0x05); o.min[2] = rd_f32(s, q * 0x06);
        o.max[0] = rd_f32(s, q * 0x07); o.max[1] = rd_f32(s, q * 0x08); o.max[2] = rd_f32(s, q * 0x09);
        uint32_t mm = 0;
        for (int k = 0; k < 3; k++)
        {
            if (rd_i32(s, q * 0x0A + 4 * k) != 0) mm |= 1u << k;        /* srf_MIN_T */
            if (rd_i32(s, q * 0x0B + 4 * k) != 0) mm |= 1u << (3 + k);  /* srf_MAX_T */
        }
        o.minmax_t = mm;

        /* a_map/a_sgn, object.cpp:2487-2497 */
        int32_t a_map[4], a_sgn[4];
        for (int k = 0; k < 4; k++)
        {
            a_map[k] = rd_i32(s, q * 0x0C + 4 * k);
            a_sgn[k] = rd_i32(s, q * 0x0D + 4 * k);
        }
        o.has_trm = a_map[3];
        o.shift   = a_sgn[3] != 0 ? 1 : 0;
        uint32_t axes = 0;
        for (int k = 0; k < 3; k++)
        {
            int32_t ax = a_map[k] / (int32_t)q;     /* 0..2 or 3..5 when shifted */
            if (ax >= 3) ax -= 3;
            axes |= (uint32_t)(ax & 3) << (2 * k);
            if (a_sgn[k] != 0) axes |= 1u << (8 + k);
        }
        o.axes = axes;

        o.smask = rd_u32(s, q * 0x0F);
        o.d_eps = rd_f32(s, q * 0x10);
        o.t_eps = rd_f32(s, q * 0x11);

        o.tci[0] = rd_f32(s, q * 0x14); o.tci[1] = rd_f32(s, q * 0x15); o.tci[2] = rd_f32(s, q * 0x16);
        o.tcj[0] = rd_f32(s, q * 0x17); o.tcj[1] = rd_f32(s, q * 0x18); o.tcj[2] = rd_f32(s, q * 0x19);
        o.tck[0] = rd_f32(s, q * 0x1A); o.tck[1] = rd_f32(s, q * 0x1B); o.tck[2] = rd_f32(s, q * 0x1C);

        o.sci[0] = rd_f32(s, q * 0x1D); o.sci[1] = rd_f32(s, q * 0x1E); o.sci[2] = rd_f32(s, q * 0x1F);
        o.sci[3] = rd_f32(s, q * 0x20);
        o.scj[0] = rd_f32(s, q * 0x21); o.scj[1] = rd_f32(s, q * 0x22); o.scj[2] = rd_f32(s, q * 0x23);

        const size_t t = q * 0x24;                  /* Q*0x240 */
        for (int k = 0; k < 4; k++) o.srf_t[k] = rd_i32(s, t + 4 * k);

        const size_t msc = t + 0x10;
        const size_t matp = msc + 0x10 * a.P;
        const size_t lstp = msc + 0x20 * a.P;

        o.conic = (int32_t)rd_cell(a, s, msc + 1 * a.ps);               /* msc_p[1] */
        uint64_t clip = rd_ptr(a, s, msc + 2 * a.ps);                    /* msc_p[2] */
        uint64_t trn  = rd_ptr(a, s, msc + 3 * a.ps);                    /* msc_p[3] */

        uint64_t m0 = rd_ptr(a, s, matp + 0 * a.ps);
        uint64_t m2 = rd_ptr(a, s, matp + 2 * a.ps);
        o.props[0] = (int32_t)rd_cell(a, s, matp + 1 * a.ps);
        o.props[1] = (int32_t)rd_cell(a, s, matp + 3 * a.ps);

        uint64_t l0 = rd_ptr(a, s, lstp + 0 * a.ps);
        uint64_t l1 = rd_ptr(a, s, lstp + 1 * a.ps);
        uint64_t l2 = rd_ptr(a, s, lstp + 2 * a.ps);
        uint64_t l3 = rd_ptr(a, s, lstp + 3 * a.ps);

        const bool real = o.srf_t[3] >= 0 && o.srf_t[3] < QR_TAG_SURFACE_MAX;

        o.clip   = real ? walk_list(clip, LIST_CLIPPERS) : QR_NULL;
        o.trnode = get_srf(trn);
        o.mat[0] = real ? get_mat(m0) : QR_NULL;
        o.mat[1] = real ? get_mat(m2) : QR_NULL;
        o.lst[0] = real ? walk_list(l0, LIST_LIGHTS)   : QR_NULL;
        o.lst[1] = real ? walk_list(l1, LIST_SURFACES) : QR_NULL;
        o.lst[2] = real ? walk_list(l2, LIST_LIGHTS)   : QR_NULL;
        o.lst[3] = real ? walk_list(l3, LIST_SURFACES) : QR_NULL;

        srf[srf_ix.at(p)] = o;
    }

    void drain()
    {
        while (!srf_todo.empty() && err.empty())
        {
            uint64_t p = srf_todo.front();
            srf_todo.pop_front();
            flatten_surface(p);
        }
    }
};

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

} /* namespace */

int qr_flatten_impl(const void *s_inf, const qr_abi_desc *abi, std::vector<uint8_t> &out, std::string &err, QrFlattenMap *map)
{
    if (s_inf == nullptr || abi == nullptr) { err = "null argument"; return QR_ERR_ARG; }
    if (abi->struct_size != sizeof(qr_abi_desc)) { err = "qr_abi_desc size mismatch"; return QR_ERR_ABI; }
    if (abi->element_bits != 32) { err = "only RT_ELEMENT=32 (fp32) builds are supported"; return QR_ERR_ABI; }
    if (abi->endian != 0) { err = "only little-endian builds are supported"; return QR_ERR_ABI; }
    if (abi->pointer_bits != 64 && abi->pointer_bits != 32) { err = "bad pointer_bits"; return QR_ERR_ABI; }
    if (abi->pointer_bits != sizeof(void *) * 8) { err = "pointer_bits does not match this process"; return QR_ERR_ABI; }
    if (abi->quads != 1 && abi->quads != 2 && abi->quads != 4 && abi->quads != 8 && abi->quads != 16)
    { err = "bad quads"; return QR_ERR_ABI; }

    static thread_local Walker tl_store;
    Walker w;
    struct Lend                     /* storage in for the call, back (emptied) on every way out */
    {
        Walker &w, &st;
        Lend(Walker &w_, Walker &st_) : w(w_), st(st_) { w.swap_storage(st); }
        ~Lend() { w.reset(); w.swap_storage(st); }
    } lend(w, tl_store);
    w.a.Q = abi->quads;
    w.a.P = abi->pointer_bits / 32;
    w.a.ps = (size_t)w.a.P * 4;
    const Abi &a = w.a;

    const uint8_t *inf = (const uint8_t *)s_inf;
    const size_t ib = (size_t)a.Q * 0x100;              /* rt_SIMD_INFO ends here, tracer.h:147 */
    auto slot = [&](int k) { return ib + (size_t)k * a.ps; };

    uint64_t p_ctx   = rd_ptr(a, inf, slot(0));
    uint64_t p_cam   = rd_ptr(a, inf, slot(1));
    uint64_t p_lst   = rd_ptr(a, inf, slot(2));
    int32_t  index   = (int32_t)rd_cell(a, inf, slot(4));
    int32_t  thnum   = (int32_t)rd_cell(a, inf, slot(5));
    int32_t  depth   = (int32_t)rd_cell(a, inf, slot(6));
    int32_t  fsaa    = (int32_t)rd_cell(a, inf, slot(7));
    int32_t  frm_w   = (int32_t)rd_cell(a, inf, slot(8));
    int32_t  frm_h   = (int32_t)rd_cell(a, inf, slot(9));
    int32_t  frm_row = (int32_t)rd_cell(a, inf, slot(10));
    int32_t  tile_w  = (int32_t)rd_cell(a, inf, slot(12));
    int32_t  tile_h  = (int32_t)rd_cell(a, inf, slot(13));
    int32_t  tls_row = (int32_t)rd_cell(a, inf, slot(14));
    uint64_t p_tiles = rd_ptr(a, inf, slot(15));
    int32_t  pt_on   = (int32_t)rd_cell(a, inf, slot(19));

    /* path-tracer mode: the snapshot records it (emission included); qr_render0 then carries the engine's seed and
     * colour planes to the device and back around the launches (qr_device.hip, dropin_pt_begin / _end) */
    w.pt = pt_on != 0;
    if (p_ctx == 0 || p_cam == 0) { err = "s_inf->ctx / cam is NULL"; return QR_ERR_ARG; }
    if (frm_w <= 0 || frm_h <= 0 || frm_w > 65536 || frm_h > 65536) { err = "bad frame size"; return QR_ERR_ARG; }
    if (tile_w <= 0 || tile_h <= 0 || tls_row <= 0 || p_tiles == 0) { err = "bad tile parameters"; return QR_ERR_ARG; }
    if (fsaa < 0 || fsaa > 2) { err = "unsupported fsaa mode"; return QR_ERR_UNSUP; }
    if (thnum <= 0 || index < 0 || index >= thnum) { err = "bad index/thnum"; return QR_ERR_ARG; }
    if (depth < 0 || depth > 64) { err = "bad depth"; return QR_ERR_ARG; }

    qr_frame f; memset(&f, 0, sizeof(f));
    const uint8_t *cam = (const uint8_t *)(uintptr_t)p_cam;
    const uint8_t *ctx = (const uint8_t *)(uintptr_t)p_ctx;
    const size_t q = (size_t)a.Q * 0x10;

    f.t_max  = rd_f32(cam, q * 0x0);
    for (int k = 0; k < 3; k++)
    {
        f.dir[k] = rd_f32(cam, q * (0x1 + k));
        f.hor[k] = rd_f32(cam, q * (0x4 + k));
        f.ver[k] = rd_f32(cam, q * (0x7 + k));
        f.amb[k] = rd_f32(cam, q * (0x11 + k));
        f.org[k] = rd_f32(ctx, q * (0x1 + k));
    }
    for (int k = 0; k < 4; k++)
    {
        f.hor_a[k] = rd_f32(cam, q * 0xA + 4 * k);     /* per-lane, period 4: engine.cpp:3480-3550 */
        f.ver_a[k] = rd_f32(cam, q * 0xB + 4 * k);
    }
    f.clamp = rd_f32(cam, q * 0xE);
    f.cmask = rd_u32(cam, q * 0xF);
    f.l_amb = rd_f32(cam, q * 0x10);
    f.t_min = rd_f32(ctx, 0);
    f.ctx_flags = (int32_t)rd_u32(ctx, q * 0x2A + 8);  /* ctx_PARAM(FLG): param[1], engine.cpp:3590 */

    f.depth = depth; f.fsaa = fsaa;
    f.frm_w = frm_w; f.frm_h = frm_h; f.frm_row = frm_row;
    f.tile_w = tile_w; f.tile_h = tile_h; f.tls_row = tls_row;
    f.tls_col = (frm_h + tile_h - 1) / tile_h;
    f.index = index; f.thnum = thnum;
    f.pt_on = pt_on != 0 ? 1 : 0;

    const bool ph = getenv("QR_VERBOSE") && atoi(getenv("QR_VERBOSE")) >= 2;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double pt = now();
    auto phase = [&](const char *n) { if (ph) { const double t = now(); fprintf(stderr, "flatten phase %-10s %.3f ms\n", n, t - pt); pt = t; } };
    /* primary lists: per-tile heads (tracer.cpp:1182-1194, 1328-1339) and inf_LST */
    std::vector<int32_t> &tiles = w.tiles;
    tiles.assign((size_t)f.tls_row * f.tls_col, QR_NULL);
    w.elm_ix.reserve(tiles.size() * 2 + 4096);          /* before anything is in it: no rehash while walking */
    w.elm.reserve(tiles.size() * 2 + 4096);
    const uint8_t *tl = (const uint8_t *)(uintptr_t)p_tiles;
    for (size_t i = 0; i < tiles.size() && w.err.empty(); i++)
    {
        uint64_t head = rd_ptr(a, tl, i * a.ps);
        tiles[i] = w.walk_list(head, LIST_SURFACES);
    }
    phase("tiles");
    f.clist = w.walk_list(p_lst, LIST_SURFACES);
    w.drain();
    phase("surfaces");
    if (!w.err.empty()) { err = w.err; return QR_ERR_ARG; }

    /* assemble blob */
    qr_header h; memset(&h, 0, sizeof(h));
    h.magic = QR_SNAPSHOT_MAGIC; h.version = QR_SNAPSHOT_VERSION;
    h.header_bytes = sizeof(qr_header);
    h.n_srf = (uint32_t)w.srf.size(); h.n_mat = (uint32_t)w.mat.size(); h.n_lgt = (uint32_t)w.lgt.size();
    h.n_elm = (uint32_t)w.elm.size(); h.n_tiles = (uint32_t)tiles.size(); h.n_texels = (uint32_t)w.texels.size();
    h.sz_frame = sizeof(qr_frame); h.sz_srf = sizeof(qr_surface); h.sz_mat = sizeof(qr_material);
    h.sz_lgt = sizeof(qr_light); h.sz_elm = sizeof(qr_elem);
    size_t off = align16(sizeof(qr_header));
    h.off_frame = (uint32_t)off;  off = align16(off + sizeof(qr_frame));
    h.off_srf = (uint32_t)off;    off = align16(off + w.srf.size() * sizeof(qr_surface));
    h.off_mat = (uint32_t)off;    off = align16(off + w.mat.size() * sizeof(qr_material));
    h.off_lgt = (uint32_t)off;    off = align16(off + w.lgt.size() * sizeof(qr_light));
    h.off_elm = (uint32_t)off;    off = align16(off + w.elm.size() * sizeof(qr_elem));
    h.off_tiles = (uint32_t)off;  off = align16(off + tiles.size() * 4);
    h.off_texels = (uint32_t)off; off = align16(off + w.texels.size() * 4);
    if (off > 0xFFFFFFFFull) { err = "snapshot exceeds 4 GiB"; return QR_ERR_NOMEM; }
    h.total_bytes = (uint32_t)off;

    out.assign(off, 0);
    memcpy(out.data(), &h, sizeof(h));
    memcpy(out.data() + h.off_frame, &f, sizeof(f));
    if (!w.srf.empty())    memcpy(out.data() + h.off_srf, w.srf.data(), w.srf.size() * sizeof(qr_surface));
    if (!w.mat.empty())    memcpy(out.data() + h.off_mat, w.mat.data(), w.mat.size() * sizeof(qr_material));
    if (!w.lgt.empty())    memcpy(out.data() + h.off_lgt, w.lgt.data(), w.lgt.size() * sizeof(qr_light));
    if (!w.elm.empty())    memcpy(out.data() + h.off_elm, w.elm.data(), w.elm.size() * sizeof(qr_elem));
    if (!tiles.empty())    memcpy(out.data() + h.off_tiles, tiles.data(), tiles.size() * 4);
    if (!w.texels.empty()) memcpy(out.data() + h.off_texels, w.texels.data(), w.texels.size() * 4);
    phase("assemble");
    if (map != nullptr)
    {
        /* which engine record became which snapshot index (qr_capture_index) */
        w.srf_ix.by_index(map->srf, w.srf.size());
        w.lgt_ix.by_index(map->lgt, w.lgt.size());
    }
    return QR_OK;
}
