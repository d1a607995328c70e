"""Synthetic scene generator: BASELINE.json config 5 ("10k quadrics") as a snapshot (include/qr_scene.h).

The reference has no such scene and its engine's list pools are quadratic in the object count
(engine.cpp:2951-2956), so this module plays the role of the engine's host side for scenes the
reference did not produce: it writes the structures render0 reads -- surfaces in the conventions of
core/engine/object.cpp (field meanings cross-checked against snapshots of the reference's own demo
scenes), materials, lights, and the surface lists:

  * one hierarchical list for everything (shadow, reflection, refraction and camera list): a
    median-split bounding-sphere tree written depth-first as nested arrays -- the reference's
    array/bounding-volume list format (tracer.cpp:3955-4054, elm_DATA = last element of the sub-list)
    is a flattened BVH with skip links;
  * per object and light a short flat shadow list of the objects that can stand between the two (what
    the engine's ssort/lsort builds with bbox_shad), the global list for the ground plane;
  * NO tile lists: the snapshot has a single whole-frame tile pointing at the camera list, and the
    backend builds per-tile lists with its GPU binning pass (QR_UPLOAD_REBIN_TILES).

Scene: N quadrics (spheres / cylinders / cones / paraboloids in turn),
centres uniform in a box above a ground plane, radii U(0.2, 1.0), materials cycling plain / metal
(reflectivity 0.5) / glass (transparency 0.5, eta 0.67 <-> 1.5), 4 lights, MT19937 seed 12345
(numpy's MT19937 raw stream equals std::mt19937's; reals are taken as (r >> 8) / 2^24).
"""
import struct

import numpy as np

MAGIC, VERSION = 0x31535251, 2
NULL = -1
SMASK = 0x80000000

P_LIGHT, P_METAL, P_GAMMA, P_FRESNEL = 0x10, 0x20, 0x40, 0x80
P_NORMAL, P_OPAQUE, P_TRANSP, P_TEXTURE = 0x100, 0x200, 0x400, 0x800
P_REFLECT, P_REFRACT, P_DIFFUSE, P_SPECULAR = 0x1000, 0x2000, 0x4000, 0x8000

TAG_PLANE, TAG_CYLINDER, TAG_SPHERE, TAG_CONE, TAG_PARABOLOID, TAG_BOUND = 0, 1, 2, 3, 4, 9


class _Rng:
    def __init__(self, seed):
        self.bg = np.random.MT19937(seed)

    def real(self, n=1):
        r = self.bg.random_raw(n).astype(np.uint64) & np.uint64(0xFFFFFFFF)
        return ((r >> np.uint64(8)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def _f(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


class _Builder:
    def __init__(self):
        self.srf = []       # list of 64-dword records (np.uint32[64])
        self.mat = []
        self.lgt = []
        self.elm = []       # [simd, data, next, kind]
        self.texels = []

    # -- records ----------------------------------------------------------------------------------
    def surface(self, tag, srf_t, pos, sci=(0, 0, 0, 0), scj=(0, 0, 0), mn=(0, 0, 0), mx=(0, 0, 0),
                minmax_t=0, conic=0, mats=(NULL, NULL), props=(0, 0), real=True):
        r = np.zeros(64, dtype=np.uint32)
        f = r.view(np.float32)
        f[0:3] = pos
        r[3] = 0xFFFFFFFF if real else 0
        f[4:7] = mn
        r[7] = minmax_t
        f[8:11] = mx
        r[11] = conic
        r[23] = 0x024                       # axes: i -> x, j -> y, k -> z, no sign flips
        f[24:28] = sci
        f[28:31] = scj
        r[31] = SMASK if real else 0
        f[32] = 1e-11                       # RT_DEPS_THRESHOLD-style constants as in the reference's snapshots
        f[33] = 1e-7
        r[34:38] = np.array(list(srf_t) + [tag], dtype=np.int32).view(np.uint32)
        r[38] = np.uint32(0xFFFFFFFF)       # clip list: none
        r[39] = np.uint32(0xFFFFFFFF)       # trnode: none
        r[40:42] = np.array(mats, dtype=np.int32).view(np.uint32)
        r[42:44] = np.array(props, dtype=np.int32).view(np.uint32)
        r[44:48] = np.uint32(0xFFFFFFFF)
        self.srf.append(r)
        return len(self.srf) - 1

    def material(self, colour, l_dff=1.0, l_spc=0.0, l_pow=16, c_rfl=0.0, c_trn=0.0, c_rfr=1.0, ext_2=0.0):
        self.texels.append(colour & 0xFFFFFF | 0xFF000000)
        r = np.zeros(32, dtype=np.uint32)
        f = r.view(np.float32)
        f[0:4] = (1.0, 1.0, 0.0, 0.0)
        r[4:7] = 0
        r[7] = len(self.texels) - 1
        r[8:10] = (0, 1)
        f[10:12] = (l_dff, l_spc)
        r[12] = l_pow
        f[13:19] = (c_rfl, c_trn, c_rfr, np.float32(c_rfr) * np.float32(c_rfr), np.float32(1.0) / np.float32(c_rfr), ext_2)
        f[19] = 255.0
        r[20] = 255
        self.mat.append(r)
        return len(self.mat) - 1

    def light(self, pos, col, l_src=1.0, a_qdr=0.0, a_lnr=0.02, a_cnt=1.0):
        r = np.zeros(16, dtype=np.float32)
        r[0] = 1.0
        r[1:4] = pos
        r[4:7] = col
        r[7] = l_src
        r[8:12] = (a_qdr, a_lnr, a_cnt, 0.0)
        self.lgt.append(r.view(np.uint32))
        return len(self.lgt) - 1

    def cell(self, simd, data=NULL, kind=0):
        self.elm.append([simd, data, NULL, kind])
        return len(self.elm) - 1

    def link(self, cells):
        for a, b in zip(cells[:-1], cells[1:]):
            self.elm[a][2] = b
        return cells[0] if cells else NULL


def make_scene(n_objects=10000, width=7680, height=4320, depth=4, seed=12345, box=100.0, gamma=False, fsaa=0,
               hierarchy=True, leaf=16, shadow_lists=True):
    """Return the snapshot bytes of the synthetic scene.  hierarchy=False writes one flat list without
    bounding-volume elements (same image, used by the tests to check that the volumes are conservative);
    shadow_lists=False gives every object the global list as shadow list (same image, slower)."""
    rng = _Rng(seed)
    b = _Builder()

    # materials: 0/1 plain (outer, inner), 2/3 metal, 4/5 glass, 6 ground
    palette = [0xD04040, 0x40B040, 0x4060D0, 0xD0C040, 0xB050C0, 0x40C0C0]
    plain = [b.material(c) for c in palette]
    metal = [b.material(c, l_dff=0.5, l_spc=0.5, l_pow=512, c_rfl=0.5, ext_2=81.0) for c in palette]
    glass_o = [b.material(c, c_trn=0.5, c_rfr=0.67) for c in palette]
    glass_i = [b.material(c, c_trn=0.5, c_rfr=1.5) for c in palette]
    ground_m = b.material(0x909090)
    inner_plain = b.material(0x808080)

    PR_PLAIN = P_DIFFUSE | P_OPAQUE | P_NORMAL
    PR_METAL = P_SPECULAR | P_DIFFUSE | P_REFLECT | P_OPAQUE | P_NORMAL | P_METAL
    PR_GLASS = P_REFRACT | P_DIFFUSE | P_NORMAL
    if gamma:
        PR_PLAIN |= P_GAMMA; PR_METAL |= P_GAMMA; PR_GLASS |= P_GAMMA

    # ground plane z = 0, clipped to a square under the box
    half = 0.75 * box
    ground = b.surface(TAG_PLANE, (1, 1, 1), (0.0, 0.0, 0.0), mn=(-half, -half, 0.0), mx=(half, half, 0.0),
                       minmax_t=0x1B, mats=(ground_m, inner_plain), props=(PR_PLAIN, PR_PLAIN))

    # objects
    n = n_objects
    cx = (rng.real(n) - np.float32(0.5)) * np.float32(box)
    cy = (rng.real(n) - np.float32(0.5)) * np.float32(box)
    cz = rng.real(n) * np.float32(box) + np.float32(1.5)
    rad = rng.real(n) * np.float32(0.8) + np.float32(0.2)
    obj = []                # (surface index, bounding-sphere centre, radius)
    for i in range(n):
        shape = i & 3
        m = i % 3
        pi = (i // 3) % len(palette)
        if m == 0:
            mats, props = (plain[pi], inner_plain), (PR_PLAIN, PR_PLAIN)
        elif m == 1:
            mats, props = (metal[pi], inner_plain), (PR_METAL, PR_PLAIN)
        else:
            mats, props = (glass_o[pi], glass_i[pi]), (PR_GLASS, PR_GLASS)
        r = float(rad[i]); p = (float(cx[i]), float(cy[i]), float(cz[i]))
        if shape == 0:      # sphere
            s = b.surface(TAG_SPHERE, (2, 3, 3), p, sci=(1, 1, 1, np.float32(r) * np.float32(r)),
                          mn=(-r, -r, -r), mx=(r, r, r), mats=mats, props=props)
            bc, br = p, r
        elif shape == 1:    # cylinder along z, height 2r
            h = 2.0 * r
            s = b.surface(TAG_CYLINDER, (2, 3, 3), p, sci=(1, 1, 0, np.float32(r) * np.float32(r)),
                          mn=(-r, -r, 0.0), mx=(r, r, h), minmax_t=0x24, mats=mats, props=props)
            bc, br = (p[0], p[1], p[2] + 0.5 * h), float(np.sqrt(r * r + 0.25 * h * h))
        elif shape == 2:    # cone, apex at pos, opening downwards, ratio 1 (radius == height)
            h = r
            s = b.surface(TAG_CONE, (2, 3, 3), p, sci=(1, 1, -1, 0), mn=(-r, -r, -h), mx=(r, r, 0.0),
                          minmax_t=0x24, conic=1, mats=mats, props=props)
            bc, br = (p[0], p[1], p[2] - 0.5 * h), float(np.sqrt(r * r + 0.25 * h * h))
        else:               # paraboloid z = (x^2 + y^2) / (2 scj), cut at height r where its radius is r
            par = 0.5 * r   # scj_z: z(r) = r^2 / (2 * 0.5 r) = r
            s = b.surface(TAG_PARABOLOID, (2, 2, 2), p, sci=(1, 1, 0, 0), scj=(0, 0, par),
                          mn=(-r, -r, 0.0), mx=(r, r, r), minmax_t=0x20, mats=mats, props=props)
            bc, br = (p[0], p[1], p[2] + 0.5 * r), float(np.sqrt(r * r + 0.25 * r * r))
        obj.append((s, np.array(bc, dtype=np.float64), br))

    # lights
    top = box + 10.0
    lights = [b.light((-0.4 * box, -0.4 * box, top), (1.0, 1.0, 1.0)),
              b.light((0.4 * box, -0.4 * box, top), (0.9, 0.9, 1.0)),
              b.light((0.4 * box, 0.4 * box, top), (1.0, 0.9, 0.9)),
              b.light((-0.4 * box, 0.4 * box, top), (0.9, 1.0, 0.9))]

    # ---- the hierarchical list ------------------------------------------------------------------
    def bound(spheres):
        """(centre, radius) of a sphere holding the given spheres (centre of their box)"""
        cs = np.array([m[0] for m in spheres]); rs = np.array([m[1] for m in spheres])
        c = 0.5 * ((cs - rs[:, None]).min(0) + (cs + rs[:, None]).max(0))
        r = float((np.linalg.norm(cs - c, axis=1) + rs).max()) * 1.0001 + 1e-4
        return c, r

    def array(member_cells, spheres):
        """prefix a bounding-volume element to a run of cells; returns (cells, (centre, radius))"""
        c, r = bound(spheres)
        bs = b.surface(TAG_BOUND, (0, 0, 0), c, sci=(1, 1, 1, np.float32(r) * np.float32(r)), real=False)
        head = b.cell(bs, data=member_cells[-1], kind=1)
        return [head] + member_cells, (c, r)

    def build(ks):
        """median-split bounding-volume tree over the objects ks, written depth-first as nested arrays:
        the reference's array list format is a flattened BVH with skip links (elm_DATA = last element)"""
        if len(ks) <= leaf:
            cells = [b.cell(obj[k][0]) for k in ks]
            sph = [(obj[k][1], obj[k][2]) for k in ks]
            return array(cells, sph) if len(ks) > 1 else (cells, sph[0])
        cs = np.array([obj[k][1] for k in ks])
        axis = int(np.argmax(cs.max(0) - cs.min(0)))
        order = [ks[i] for i in np.argsort(cs[:, axis], kind="stable")]
        l_cells, l_sph = build(order[:len(order) // 2])
        r_cells, r_sph = build(order[len(order) // 2:])
        return array(l_cells + r_cells, [l_sph, r_sph])

    top_cells = [b.cell(ground)]
    if hierarchy and obj:
        top_cells += build(list(range(len(obj))))[0]
    else:
        top_cells += [b.cell(o[0]) for o in obj]
    glist = b.link(top_cells)

    # light lists.  Default: one light list for everything, every light's shadow list is the global list.
    # Which side of a surface a light is entered on is the engine's rule (RT_OPTS_2SIDED: lsort -> bbox_side -> clip_side,
    # engine.cpp:2503-2533, rtgeom.cpp:939-995; the same rule as light_sides in csrc/qr_compile.cpp): the sign of the
    # quadric form at the light (margin 1e-4); a convex surface seen from inside shows its inner side only; from outside
    # the outer side only, unless the surface is concave-capable with holes or the light stands outside its clip box.
    lpos32 = [b.lgt[l].view(np.float32)[1:4].copy() for l in lights]

    def sides(r, lp):
        f = r.view(np.float32)
        tag = int(r[37].view(np.int32)) if hasattr(r[37], "view") else int(np.int32(r[37]))
        loc = (lp - f[0:3]).astype(np.float32)
        if tag == 0:
            k = (int(r[23]) >> 4) & 3
            d = -loc[k] if (int(r[23]) >> 10) & 1 else loc[k]
        else:
            sci, scj = f[24:28], f[28:31]
            dcj = np.float32(loc[0] * (scj[0] + scj[0]) + loc[1] * (scj[1] + scj[1]) + loc[2] * (scj[2] + scj[2]))
            dci = np.float32(loc[0] * loc[0] * sci[0] + loc[1] * loc[1] * sci[1] + loc[2] * loc[2] * sci[2])
            d = np.float32(dci - dcj - sci[3])
        c = 2 if d > 1e-4 else (0 if d >= -1e-4 else 1)
        if c == 0:
            return 3
        if tag == 0:
            return c
        if tag not in (3, 5, 7, 8) and c == 1:
            return c
        mm = int(r[7]) & 63
        if mm == 0:
            return c
        for a in range(3):
            cmin = f[4 + a] + f[a] if mm & (1 << a) else -np.inf
            cmax = f[8 + a] + f[a] if mm & (1 << (3 + a)) else np.inf
            if lp[a] - 1e-4 <= cmin or lp[a] + 1e-4 >= cmax:
                return 3
        return c

    def light_lists(r, cells):
        """cells: one (light, shadow list) per light -> heads of the outer and the inner light list of surface r"""
        out = []
        for want in (2, 1):
            key = tuple((l, sh) for (l, sh), lp in zip(cells, lpos32) if sides(r, lp) & want)
            if key not in shared:                     # equal lists are one list (every light on the global shadow list: shared by all)
                shared[key] = b.link([b.cell(l, data=sh) for l, sh in key]) if key else NULL
            out.append(shared[key])
        return out

    shared = {}

    for r in b.srf:
        if int(r[37]) < TAG_BOUND:
            lo, li_ = light_lists(r, [(l, glist) for l in lights])
            r[44:48] = np.array([lo, glist, li_, glist], dtype=np.int32).view(np.uint32)
    if shadow_lists and obj:
        # Per object and light, the engine's kind of shadow list (ssort/lsort + bbox_shad, engine.cpp:2134-2753):
        # only the objects that can stand between the light and this object -- those whose bounding sphere
        # meets the cone from the light over the object's bounding sphere (the object itself included: an
        # open quadric can shadow its own inside).  Flat lists: they are short.  The ground plane keeps
        # the global hierarchical list.
        cen = np.array([o[1] for o in obj]); rad_b = np.array([o[2] for o in obj])
        lpos = np.array([b.lgt[l].view(np.float32)[1:4] for l in lights], dtype=np.float64)
        heads = np.full((len(obj), len(lights)), NULL, dtype=np.int64)
        for li in range(len(lights)):
            P = cen - lpos[li]                                   # light -> every object
            P2 = (P * P).sum(1)
            for c0 in range(0, len(obj), 256):                   # 256 cone axes at a time against all objects
                d = P[c0:c0 + 256]; D2 = P2[c0:c0 + 256]; D = np.sqrt(D2); r = rad_b[c0:c0 + 256]
                Pd = d @ P.T                                     # [axis, object]
                t = Pd / D2[:, None]
                tc = np.clip(t, 0.0, 1.0)
                dist2 = np.maximum(P2[None, :] - 2.0 * tc * Pd + tc * tc * D2[:, None], 0.0)
                reach = (rad_b[None, :] + r[:, None] * tc) * 1.05 + 1e-3
                hit = (dist2 <= reach * reach) & (t <= 1.0 + (r[:, None] + rad_b[None, :]) / D[:, None] * 1.05 + 1e-3) \
                    & (t >= -rad_b[None, :] / D[:, None] - 1e-3)
                hit[np.arange(len(d)), np.arange(c0, c0 + len(d))] = True
                for j in range(len(d)):
                    heads[c0 + j, li] = b.link([b.cell(obj[y][0]) for y in np.nonzero(hit[j])[0]])
        for k, (s, c, r) in enumerate(obj):
            lo, li_ = light_lists(b.srf[s], [(l, int(heads[k, li])) for li, l in enumerate(lights)])
            b.srf[s][44] = np.uint32(lo & 0xFFFFFFFF); b.srf[s][46] = np.uint32(li_ & 0xFFFFFFFF)

    # ---- camera: outside a corner of the box, looking at its centre ------------------------------
    eye = np.array([-0.95 * box, -1.25 * box, 0.9 * box + 1.5])
    target = np.array([0.0, 0.0, 0.45 * box])
    fwd = target - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pov = 1.0
    step = 1.0 / width                                            # image plane one unit wide at distance pov
    hor = right * step; ver = down * step
    tl = fwd * pov - hor * (0.5 * width) - ver * (0.5 * height)

    fr = np.zeros(49, dtype=np.uint32)                             # qr_frame: 41 used dwords + pad[8]
    ff = fr.view(np.float32)
    ff[0] = np.finfo(np.float32).max
    ff[1:4] = tl; ff[4:7] = hor; ff[7:10] = ver
    if fsaa == 2:                                                  # engine.cpp:3525-3546 style 4x pattern
        ff[10:14] = (-0.25 - 0.08, 0.25 - 0.08, -0.25 + 0.08, 0.25 + 0.08)
        ff[14:18] = (-0.25 + 0.08, -0.25 - 0.08, 0.25 + 0.08, 0.25 - 0.08)
    ff[18] = 255.0; fr[19] = 255
    ff[20] = 0.15; ff[21:24] = (0.15, 0.15, 0.15)
    ff[24] = pov; ff[25:28] = eye
    fr[28] = P_GAMMA if gamma else 0
    fr[29] = depth; fr[30] = fsaa
    fr[31:34] = (width, height, width)
    fr[34:38] = (width, height, 1, 1)                              # a single whole-frame tile
    fr[38] = glist; fr[39] = 0; fr[40] = 1

    # ---- serialise ------------------------------------------------------------------------------
    def pad16(x):
        return (x + 15) & ~15
    srf = np.array(b.srf, dtype=np.uint32); mat = np.array(b.mat, dtype=np.uint32)
    lgt = np.array(b.lgt, dtype=np.uint32); elm = np.array(b.elm, dtype=np.int32)
    tiles = np.array([glist], dtype=np.int32); tex = np.array(b.texels, dtype=np.uint32)
    o_frame = 128
    o_srf = pad16(o_frame + fr.nbytes); o_mat = pad16(o_srf + srf.nbytes); o_lgt = pad16(o_mat + mat.nbytes)
    o_elm = pad16(o_lgt + lgt.nbytes); o_tiles = pad16(o_elm + elm.nbytes); o_tex = pad16(o_tiles + tiles.nbytes)
    total = pad16(o_tex + tex.nbytes)
    hdr = struct.pack("<4I6I7I5I10I", MAGIC, VERSION, total, 128,
                      len(srf), len(mat), len(lgt), len(elm), 1, len(tex),
                      o_frame, o_srf, o_mat, o_lgt, o_elm, o_tiles, o_tex,
                      fr.nbytes, 256, 128, 64, 16, *([0] * 10))
    blob = bytearray(total)
    blob[0:128] = hdr
    for off, arr in ((o_frame, fr), (o_srf, srf), (o_mat, mat), (o_lgt, lgt), (o_elm, elm), (o_tiles, tiles), (o_tex, tex)):
        blob[off:off + arr.nbytes] = arr.tobytes()
    return bytes(blob)
