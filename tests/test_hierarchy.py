"""Object hierarchy and animators (include/qr_hierarchy.h, SURVEY.md 8f row 4: rt_Object::update_matrix, object.cpp:221-389).

Fixtures: tests/golden/tree/*.json.gz, dumped from the unmodified reference engine after a frame (oracle/ref_driver.cpp
--tree, tests/golden/make_tree_golden.py): per object the inputs of the hierarchical update and the engine's results.
The bar is bit-exact fp32: matrices, transform nodes and flags per object; every transform field of the snapshot of the
same frame; and, end to end, the reference's frame at another animation time from a snapshot patched by this module."""
import gzip
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, MANIFEST, load_blob, load_frame

TREE = os.path.join(GOLDEN, "tree")
CASES = sorted(n[:-8] for n in os.listdir(TREE) if n.endswith(".json.gz") and n[:-8] in MANIFEST)
# transforms fuzzed inside the reference engine (ref_driver --jitter): tree + that run's snapshot (texels zeroed)
FUZZ = sorted(n[:-8] for n in os.listdir(TREE) if n.startswith("fuzz_") and n.endswith(".json.gz"))
# pairs of one scene "a moment later" (ref_driver --shift A / B: two thirds of the objects moved, arrays with bounding volumes
# among them): (state the snapshot is patched from, state it is patched to -- tree, the engine's snapshot and frame)
MOVED = [("moved_demo02_s1", "moved_demo02_s2"), ("moved_demo03_s3", "moved_demo03_s4"), ("moved_demo01_s5", "moved_demo01_s6"),
         ("moved_test14_s3", "moved_test14_s4"), ("moved_test16_s3", "moved_test16_s4"), ("moved_demo03_swarm_s7", "moved_demo03_swarm_s8")]
MOVED_ALL = sorted(set(n for p in MOVED for n in p))

# (snapshot + tree the scene is patched from, tree of the target time, where the reference's frame of that time is)
ANIMATED = [
    ("demo01_160_t12345", "demo01_160_t2500", "tree"),
    ("demo01_160_gf_t5000", "demo01_160_gf_t2500", "tree"),
    ("demo02_160_gf_t5000", "demo02_160_t7777_gf", "golden"),
    ("demo03_160", "demo03_160_t3000", "tree"),
    # round 4: the SET of transform nodes changes between the two times -- the light's array of demo scenes 1 and 2 leaves its
    # right angle at t > 0 and becomes the transform node of its bulb (a record and a list element are created), or returns
    ("demo01_160", "demo01_160_t2500", "tree"),
    ("demo01_160", "demo01_160_t12345", "golden"),
    ("demo01_160_t12345", "demo01_160", "golden"),
    ("demo02_160", "demo02_160_t4000", "tree"),
    ("demo03_160", "demo03_160_t7000", "tree"),
]


def _f32(words):
    return np.array([int(x, 16) for x in words], dtype=np.uint32).view(np.float32)


def load_tree(qr, name):
    with open(os.path.join(TREE, name + ".json.gz"), "rb") as f:
        t = json.loads(gzip.decompress(f.read()))
    return t, nodes_from_tree(qr, t)


def nodes_from_tree(qr, t):
    nodes = np.zeros(len(t["nodes"]), dtype=qr.node_dtype())
    for i, n in enumerate(t["nodes"]):
        r = nodes[i]
        r["parent"], r["tag"] = n["parent"], n["tag"]
        r["scl"], r["rot"], r["pos"], r["shape"] = _f32(n["scl"]), _f32(n["rot"]), _f32(n["pos"]), _f32(n["shape"])
        r["srf"], r["inb"], r["bvb"], r["lgt"] = n.get("srf", -1), n.get("inb", -1), n.get("bvb", -1), n.get("lgt", -1)
        r["anim"] = -1
        r["bvnode"], r["nverts"] = n.get("bvnode", -1), n.get("verts_num", 0)
        if "lmin" in n:
            r["lmin"], r["lmax"] = _f32(n["lmin"]), _f32(n["lmax"])
        if "tex" in n:
            r["tex"], r["has_tex"] = _f32(n["tex"]), 1
        if "pov" in n:
            r["pov"] = _f32(n["pov"])[0]
    return nodes


def tree_frame(name):
    with open(os.path.join(TREE, name + ".frame.npy.gz"), "rb") as f:
        return np.load(io.BytesIO(gzip.decompress(f.read())))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _fuzz_blob(name):
    with open(os.path.join(TREE, name + ".qrs.gz"), "rb") as f:
        return gzip.decompress(f.read())


@pytest.mark.parametrize("name", CASES + FUZZ)
def test_update_and_fields_match_the_engine(qr, name):
    """Per object: matrix (bit for bit), transform node, transform flags.  Then the snapshot of the same frame: writing
    the fields the nodes imply (position, inverse matrices, axis maps, quadric coefficients, light positions, camera
    vectors) must change nothing.  The stock scenes, and scenes whose every object got another transform inside the
    engine (right-angle and arbitrary rotations, negative and non-unit scalers; 300 such variants agreed when the
    fixtures were made, nine are kept)."""
    t, nodes = load_tree(qr, name)
    st = qr.hierarchy_update(nodes, t["opts"])
    for i, n in enumerate(t["nodes"]):
        assert (bits(st[i]["mtx"]) == bits(_f32(n["mtx"]))).all(), (i, n["tag"])
        assert (st[i]["trnode"], st[i]["obj_has_trm"], st[i]["mtx_has_trm"]) == (n["trnode"], n["obj_has_trm"], n["mtx_has_trm"]), i
    blob = _fuzz_blob(name) if name in FUZZ else load_blob(name)
    assert qr.hierarchy_apply(blob, nodes, t["opts"], camera=t["camera"]) == blob
    assert qr.hierarchy_apply(blob, nodes, t["opts"], camera=t["camera"], base=nodes) == blob
    # with the bounds update on top (clip boxes of every surface, records of the arrays' bounding volumes): still nothing
    assert qr.hierarchy_apply(blob, nodes, t["opts"], camera=t["camera"], base=nodes, flags=qr.HIER_BOUNDS) == blob


@pytest.mark.parametrize("name", CASES + FUZZ + MOVED_ALL)
def test_bounds_match_the_engine(qr, name):
    """Bounding and clipping boxes (qr_hierarchy_bounds: rt_Surface::update_minmax / update_bounds, rt_Array::update_bounds,
    object.cpp:1830-2318, 2534-2845), bit for bit: every surface's bounding box, clipping box, centre and radius of the
    box's corners; every array's inbox, bvbox and trbox with their radii."""
    t, nodes = load_tree(qr, name)
    blob = _fuzz_blob(name) if (name in FUZZ or name in MOVED_ALL) else load_blob(name)
    b = qr.hierarchy_bounds(blob, nodes, t["opts"])
    n_srf = n_arr = 0
    for i, n in enumerate(t["nodes"]):
        if "bmin" in n:
            n_srf += 1
            for k in ("bmin", "bmax", "cmin", "cmax"):
                assert (bits(b[i][k]) == bits(_f32(n[k]))).all(), (i, n["tag"], k)
            assert b[i]["nverts"] == n["verts_num"]
            if n["verts_num"]:
                assert (bits(b[i]["mid"]) == bits(_f32(n["mid"]))).all() and bits(b[i]["rad"]) == bits(_f32(n["rad"]))[0], i
        if "inbox_min" in n:
            n_arr += 1
            for k, key in (("inmin", "inbox_min"), ("inmax", "inbox_max"), ("bmin", "bvbox_min"), ("bmax", "bvbox_max"),
                           ("trmin", "trbox_min"), ("trmax", "trbox_max")):
                assert (bits(b[i][k]) == bits(_f32(n[key]))).all(), (i, k)
            for k, key in (("inrad", "inbox_rad"), ("rad", "bvbox_rad"), ("trrad", "trbox_rad")):
                assert bits(b[i][k]) == bits(_f32(n[key]))[0], (i, k)
    assert n_srf > 0 and n_arr > 0


def test_bounds_reject_a_broken_node_table(qr):
    """record indices outside the snapshot, a bounding-volume node that is no array in front of the node: refused, not read"""
    t, nodes = load_tree(qr, "demo02_160")
    blob = load_blob("demo02_160")
    srf = [i for i, n in enumerate(t["nodes"]) if "bmin" in n]
    bad = nodes.copy(); bad[srf[0]]["srf"] = 1 << 20
    with pytest.raises(qr.QrError):
        qr.hierarchy_bounds(blob, bad, t["opts"])
    bad = nodes.copy(); bad[srf[0]]["bvnode"] = len(nodes) - 1
    with pytest.raises(qr.QrError):
        qr.hierarchy_bounds(blob, bad, t["opts"])
    bad = nodes.copy(); bad[srf[3]]["bvnode"] = srf[0]                  # a surface, not an array
    with pytest.raises(qr.QrError):
        qr.hierarchy_bounds(blob, bad, t["opts"])
    with pytest.raises(qr.QrError):
        qr.hierarchy_bounds(blob[:4096], nodes, t["opts"])              # truncated snapshot


# words of a qr_surface record the hierarchy owns (include/qr_scene.h): everything but list heads (clip 38, lst 44-47),
# the transform node's and the materials' indices (39-41: numbered per snapshot) and padding
_OWNED = [w for w in range(44) if w not in (38, 39, 40, 41)]


def _records(blob):
    import struct
    h = struct.unpack_from("<26I", blob, 0)
    return np.frombuffer(blob, dtype=np.uint32, count=h[4] * 64, offset=h[11]).reshape(h[4], 64)


def _moved_patch(qr, a, b, flags):
    ta, na = load_tree(qr, a)
    tb, nb = load_tree(qr, b)
    assert len(na) == len(nb) and (na["parent"] == nb["parent"]).all() and (na["tag"] == nb["tag"]).all()
    nxt = nb.copy()
    for k in ("srf", "inb", "bvb", "lgt"):
        nxt[k] = na[k]                      # the records of the snapshot that is patched
    return ta, na, tb, nb, qr.hierarchy_apply(_fuzz_blob(a), nxt, tb["opts"], camera=tb["camera"], base=na, flags=flags)


@pytest.mark.parametrize("a,b", MOVED)
def test_moved_bounding_volumes_give_the_engines_records(qr, a, b):
    """Arrays with bounding volumes move: without QR_HIER_BOUNDS the update is refused; with it EVERY field of every
    surface, array and bounding-volume record of the patched snapshot equals the engine's own snapshot of the target state
    (positions, clip boxes and which sides clip, quadric coefficients, maps; the volumes' centres and coefficients)."""
    with pytest.raises(qr.QrError, match="bounding volume"):
        _moved_patch(qr, a, b, 0)
    ta, na, tb, nb, out = _moved_patch(qr, a, b, qr.HIER_BOUNDS)
    got, want = _records(out), _records(_fuzz_blob(b))
    n = 0
    for i in range(len(na)):
        for k in ("srf", "inb", "bvb"):
            ia, ib = int(na[i][k]), int(nb[i][k])
            if ia < 0 or ib < 0:
                assert ia < 0 and ib < 0
                continue
            assert (got[ia, _OWNED] == want[ib, _OWNED]).all(), (i, k, [w for w in _OWNED if got[ia, w] != want[ib, w]])
            n += 1
    assert n > 10
    moved_volumes = sum(1 for i in range(len(na)) if na[i]["tag"] == -1 and na[i]["inb"] >= 0
                        and (got[int(na[i]["inb"])] != _records(_fuzz_blob(a))[int(na[i]["inb"])]).any())
    assert moved_volumes > 0, "no bounding volume changed: the pair does not test anything"


@pytest.mark.parametrize("a,b", MOVED)
def test_moved_scene_renders_to_the_reference_frame(qr, oracle, a, b):
    """... and end to end: patched snapshot -> rebuilt lists -> the oracle renders the reference's frame of the target state."""
    _, _, _, _, out = _moved_patch(qr, a, b, qr.HIER_BOUNDS | qr.HIER_RESET_TILES)
    frame, _, _ = oracle.render(qr.build_lists(out), threads=8)
    assert (frame == (tree_frame(b) & 0xFFFFFF)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("a,b", MOVED)
def test_gpu_moved_scene_renders_to_the_reference_frame(qr, a, b):
    _, _, _, _, out = _moved_patch(qr, a, b, qr.HIER_BOUNDS | qr.HIER_RESET_TILES)
    sc = qr.Scene(qr.build_lists(out), rebin_tiles=True)
    frame = sc.render().cpu().numpy().view(np.uint32) & 0xFFFFFF
    assert (frame == (tree_frame(b) & 0xFFFFFF)).all()


def _animators(t):
    """The demo scenes' animators as data: the array above the camera turns 1 degree per 50 ms (demo01) or swings
    15 degrees with sin(t / 1500) (demo03); the array above a light turns 7 degrees per 50 ms (demo01, demo02)."""
    slots, anim = [], {}
    for i, n in enumerate(t["nodes"]):
        if n["anim"]:
            kids = [c["tag"] for c in t["nodes"] if c["parent"] == i]
            if 100 in kids:
                slots.append(("swing", 2, 15.0, 1500.0) if t["scene"] == "demo03" else ("spin", 2, 1.0))
            else:
                slots.append(("spin", 2, 7.0))
            anim[i] = len(slots) - 1
    return slots, anim


@pytest.mark.parametrize("start,target", [("demo01_160", "demo01_160_t2500"), ("demo01_160", "demo01_160_gf_t5000"),
                                          ("demo01_160", "demo01_160_t12345"), ("demo02_160", "demo02_160_t7777_gf"),
                                          ("demo03_160", "demo03_160_t3000"), ("demo03_160", "demo03_160_t9999_aa4")])
def test_animators_move_the_nodes_like_the_engines(qr, start, target):
    t0, nodes = load_tree(qr, start)
    t1, want = load_tree(qr, target)
    slots, anim = _animators(t0)
    assert anim, "the scene has animators"
    for i, k in anim.items():
        nodes[i]["anim"] = k
    times = np.full(len(nodes), -1, dtype=np.int64)
    qr.hierarchy_animate(nodes, t1["time"], times, slots)
    for f in ("scl", "rot", "pos"):
        assert (bits(nodes[f]) == bits(want[f])).all(), f
    assert (times == t1["time"]).all()
    before = nodes.copy()
    qr.hierarchy_animate(nodes, t1["time"], times, slots)          # same time again: animators are not called twice
    assert nodes.tobytes() == before.tobytes()
    # a Python callable in a slot sees (time, last_time, the nine floats)
    seen = []
    fresh = load_tree(qr, start)[1]
    for i in anim:
        fresh[i]["anim"] = 0

    def cb(time, last, trm):
        seen.append((time, last))
        trm[3 + 2] += 1.0

    qr.hierarchy_animate(fresh, 40, np.full(len(fresh), 7, dtype=np.int64), [cb])
    assert seen == [(40, 7)] * len(anim)


def _patched(qr, base_name, target_name):
    tb, base = load_tree(qr, base_name)
    tt, tgt = load_tree(qr, target_name)
    assert len(base) == len(tgt) and (base["parent"] == tgt["parent"]).all() and (base["tag"] == tgt["tag"]).all()
    nxt = base.copy()                       # the base snapshot's record indices, the target time's transforms
    for f in ("scl", "rot", "pos"):
        nxt[f] = tgt[f]
    blob = qr.hierarchy_apply(load_blob(base_name), nxt, tb["opts"], camera=tb["camera"], base=base,
                              flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
    return qr.build_lists(blob)


def _target_frame(target_name, where):
    return (tree_frame(target_name) if where == "tree" else load_frame(target_name)) & 0xFFFFFF


@pytest.mark.parametrize("base_name,target_name,where", ANIMATED)
def test_animated_snapshot_gives_the_reference_frame_of_that_time(qr, oracle, base_name, target_name, where):
    """Snapshot of one time + the hierarchy at another time -> apply -> rebuilt lists -> the oracle renders the
    reference's frame of that other time, pixel for pixel."""
    built = _patched(qr, base_name, target_name)
    frame, _, _ = oracle.render(built, threads=8)
    assert (frame == _target_frame(target_name, where)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("base_name,target_name,where", ANIMATED)
def test_gpu_animated_snapshot_gives_the_reference_frame_of_that_time(qr, base_name, target_name, where):
    built = _patched(qr, base_name, target_name)
    sc = qr.Scene(built, rebin_tiles=True)
    frame = sc.render().cpu().numpy().view(np.uint32) & 0xFFFFFF
    assert (frame == _target_frame(target_name, where)).all()


def _shuffled_global_list(blob, seed):
    """The snapshot with its global list in another order: groups (array element + members) stay together and move as one,
    members are permuted inside their group; every tile gets the list."""
    import random
    import struct
    rnd = random.Random(seed)
    b = bytearray(blob)
    h = struct.unpack_from("<4I6I7I5I", b, 0)
    _, E, c = _snapshot_view(blob)
    items, e = [], c
    while e != -1:
        simd, data, nxt, kind = (int(x) for x in E[e])
        if data != -1:
            mem, m = [], nxt
            while True:
                mem.append(m)
                if m == data:
                    break
                m = int(E[m][2])
            rnd.shuffle(mem)
            items.append([e] + mem)
            e = int(E[data][2])
        else:
            items.append([e])
            e = nxt
    rnd.shuffle(items)
    flat = [x for it in items for x in it]
    for a, nx in zip(flat, flat[1:] + [-1]):
        struct.pack_into("<i", b, h[14] + 16 * a + 8, nx)
    for it in items:
        if len(it) > 1:
            struct.pack_into("<i", b, h[14] + 16 * it[0] + 4, it[-1])
    struct.pack_into("<i", b, h[10] + 4 * 38, flat[0])
    for k in range(h[8]):
        struct.pack_into("<i", b, h[15] + 4 * k, flat[0])
    return bytes(b)


@pytest.mark.parametrize("name", ["demo01_160", "demo02_160", "demo03_160", "test03_160", "test13_160", "test16_160"])
def test_the_order_of_the_global_list_decides_no_pixel(qr, oracle, name):
    """Why qr_hierarchy_apply may keep the previous order where the engine would re-sort by view order (include/qr_hierarchy.h):
    the list's order decides exact depth ties only.  The reference's own capture with its global list shuffled (two seeds),
    lists rebuilt from it, renders to the reference's frame pixel for pixel."""
    blob = load_blob(name)
    want = load_frame(name) & 0xFFFFFF
    for seed in (1, 2):
        frame, _, _ = oracle.render(qr.build_lists(_shuffled_global_list(blob, seed)), threads=4)
        assert np.array_equal(frame, want), seed


def test_numpy_mirrors_have_the_c_layout(qr, tmp_path):
    """quadray-engine_amd.node_dtype / node_state_dtype / node_bounds_dtype against include/qr_hierarchy.h as gcc lays it out."""
    import shutil
    import subprocess
    from conftest import ROOT
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "qr_hierarchy.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(qr_node), sizeof(qr_node_state), sizeof(qr_node_bounds),'
                   ' offsetof(qr_node, tex), offsetof(qr_node, has_tex), offsetof(qr_node, lmin));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    nd = qr.node_dtype()
    assert got == [nd.itemsize, qr.node_state_dtype().itemsize, qr.node_bounds_dtype().itemsize,
                   nd.fields["tex"][1], nd.fields["has_tex"][1], nd.fields["lmin"][1]]


def test_a_patched_snapshot_is_patched_again_frame_after_frame(qr, oracle):
    """An animation loop that keeps ONE snapshot: demo scene 1 at t = 0 -> 2.5 s -> 12.345 s -> 0, every step patching the
    previous step's output (not the t = 0 capture).  The light's array becomes a transform node in the first step -- its record
    is n_srf + 0, entered into the table with hierarchy_records_after_apply -- stays one, and stops being one in the last; every
    step renders to the reference's frame of its time, and the last snapshot has the first one's records and list again."""
    t0, table = load_tree(qr, "demo01_160")
    blob = load_blob("demo01_160")
    flags = qr.HIER_RESET_TILES | qr.HIER_BOUNDS
    for target, where in (("demo01_160_t2500", "tree"), ("demo01_160_t12345", "golden"), ("demo01_160", "golden")):
        _, tgt = load_tree(qr, target)
        nxt = table.copy()
        for f in ("scl", "rot", "pos"):
            nxt[f] = tgt[f]
        patched = qr.hierarchy_apply(blob, nxt, t0["opts"], camera=t0["camera"], base=table, flags=flags)
        frame, _, _ = oracle.render(qr.build_lists(patched), threads=4)
        assert (frame == _target_frame(target, where)).all(), target
        table = qr.hierarchy_records_after_apply(blob, nxt, t0["opts"])
        blob = patched
    S0, _, c0 = _snapshot_view(load_blob("demo01_160"))
    S1, _, c1 = _snapshot_view(blob)
    assert len(S1) == len(S0) + 1                                       # the light array's record stays behind, unlinked
    assert (S1[:len(S0), :38] == S0[:, :38]).all()
    assert _list_shape(blob, c1) == _list_shape(load_blob("demo01_160"), c0)


def _snapshot_view(blob):
    import struct
    f = struct.unpack_from("<4I6I7I5I", blob, 0)
    n_srf, n_elm, o_frame, o_srf, o_elm = f[4], f[7], f[10], f[11], f[14]
    S = np.frombuffer(blob, dtype=np.uint32, count=n_srf * 64, offset=o_srf).reshape(n_srf, 64)
    E = np.frombuffer(blob, dtype=np.int32, count=n_elm * 4, offset=o_elm).reshape(n_elm, 4)
    return S, E, int(np.frombuffer(blob, dtype=np.int32, count=64, offset=o_frame)[38])


@pytest.mark.parametrize("base_name,target_name", [("demo01_160", "demo01_160_t12345"), ("demo01_160_t12345", "demo01_160")])
def test_changing_set_of_transform_nodes_gives_the_engines_records_and_list_order(qr, base_name, target_name):
    """An array that starts (or stops) being a transform node: qr_hierarchy_apply creates (drops) its record and its element
    of the global list.  Against the ENGINE's snapshot of the target time, node by node: every transform field of every record
    -- the new transform node's included -- and the camera list member for member, array elements and kinds included."""
    tb, base = load_tree(qr, base_name)
    _, tgt = load_tree(qr, target_name)
    nxt = base.copy()
    for f in ("scl", "rot", "pos"):
        nxt[f] = tgt[f]
    # QR_HIER_BOUNDS: the members' boxes are expressed in the new transform node's space (min / max relative to the position)
    patched = qr.hierarchy_apply(load_blob(base_name), nxt, tb["opts"], camera=tb["camera"], base=base,
                                 flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
    Sp, Ep, cp = _snapshot_view(patched)
    Se, Ee, ce = _snapshot_view(load_blob(target_name))
    rec_p = {i: int(base[i]["srf"]) for i in range(len(base)) if base[i]["srf"] >= 0}
    born = [i for i in range(len(tgt)) if tgt[i]["srf"] >= 0 and base[i]["srf"] < 0]
    assert len(born) <= 1
    for i in born:
        rec_p[i] = len(Sp) - 1                      # the record apply appended
    assert len(Sp) == _snapshot_view(load_blob(base_name))[0].shape[0] + len(born)
    # qr_surface words: pos 0-2, c_def 3, min 4-6, minmax_t 7, max 8-10, conic 11, tci 12-14, has_trm 15, tcj 16-18, shift 19,
    # tck 20-22, axes 23, sci 24-27, scj 28-30, smask 31, d_eps 32, t_eps 33, srf_t 34-37, (clip 38), trnode 39
    words = list(range(0, 38))
    node_of_e = {int(tgt[i]["srf"]): i for i in range(len(tgt)) if tgt[i]["srf"] >= 0}
    node_of_p = {r: i for i, r in rec_p.items()}
    for i in range(len(tgt)):
        if tgt[i]["srf"] < 0 or i not in rec_p:
            continue
        a, b = Sp[rec_p[i]], Se[int(tgt[i]["srf"])]
        assert (a[words] == b[words]).all(), (i, [w for w in words if a[w] != b[w]])
        ta, tb_ = int(a.view(np.int32)[39]), int(b.view(np.int32)[39])
        assert (ta < 0) == (tb_ < 0) and (ta < 0 or node_of_p[ta] == node_of_e[tb_]), i      # the same transform node

    def chain(S, E, head, node_of):
        out, e = [], head
        while e != -1:
            simd, data, nxt_, kind = (int(x) for x in E[e])
            out.append((node_of.get(simd, ("record", simd)), data != -1, kind))
            e = nxt_
        return out
    ours, engines = chain(Sp, Ep, cp, node_of_p), chain(Se, Ee, ce, node_of_e)
    assert ours == engines
    # the array element's run ends at the same member
    def last_of(S, E, head, node_of):
        e = head
        while e != -1:
            if E[e][1] != -1:
                return node_of[int(E[int(E[e][1])][0])]
            e = int(E[e][2])
        return None
    assert last_of(Sp, Ep, cp, node_of_p) == last_of(Se, Ee, ce, node_of_e)


# Jittered captures of the reference's test scenes (tests/golden/make_tree_golden.py: every node's transform drawn anew, same
# options): between the two times surfaces start and stop being their OWN transform node (a right angle <-> any angle), and
# planes take their axis scalers in and out of the matrix, which moves the texture scale of their materials.
SELF_NODES = [("test05_160", "test05_160_j1"), ("test05_160_j1", "test05_160"), ("test08_160", "test08_160_j4"), ("test08_160_j4", "test08_160"),
              ("test11_160", "test11_160_j7"), ("test11_160_j7", "test11_160"), ("test15_160", "test15_160_j9"),
              ("test15_160_j9", "test15_160"), ("test18_160", "test18_160_j18"), ("test18_160_j18", "test18_160"),
              ("test07_160_gf", "test07_160_j3"), ("test07_160_j3", "test07_160_gf"), ("test12_160_noopt", "test12_160_j8"),
              ("test12_160_j8", "test12_160_noopt"), ("test17_160", "test17_160_j18"), ("test17_160_j18", "test17_160")]


def _apply_towards(qr, base_name, target_name, with_tex=True):
    tb, base = load_tree(qr, base_name)
    tt, tgt = load_tree(qr, target_name)
    assert tb["opts"] == tt["opts"] and len(base) == len(tgt)
    nxt = base.copy()
    for f in ("scl", "rot", "pos"):
        nxt[f] = tgt[f]
    if not with_tex:
        nxt["has_tex"] = 0
    patched = qr.hierarchy_apply(load_blob(base_name), nxt, tb["opts"], camera=tb["camera"], base=base,
                                 flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
    return patched, base, tgt, qr.hierarchy_update(base, tb["opts"]), qr.hierarchy_update(nxt, tb["opts"])


@pytest.mark.parametrize("base_name,target_name", SELF_NODES)
def test_surfaces_entering_and_leaving_their_own_transform_node(qr, oracle, base_name, target_name):
    """A surface that becomes / stops being its own transform node (object.cpp:445-519: a non-trivial rotation or, without
    RT_OPTS_FSCALE folding, scale): no list element changes, the record carries the matrix itself.  Against the engine's
    snapshot of the target: every transform word of every record, the texture scale and offset of the planes' materials
    (rt_Plane::update_fields, object.cpp:2893-2938), and the frame pixel for pixel after the lists are rebuilt."""
    patched, base, tgt, st0, st1 = _apply_towards(qr, base_name, target_name)
    changed = [i for i in range(len(base)) if 0 <= base[i]["tag"] < 9 and base[i]["srf"] >= 0
               and (int(st0[i]["trnode"]) == i) != (int(st1[i]["trnode"]) == i)]
    assert changed, "the pair does not exercise the case"
    Sp, Ep, cp = _snapshot_view(patched)
    target = load_blob(target_name)
    Se, Ee, ce = _snapshot_view(target)
    for i in range(len(tgt)):
        if tgt[i]["srf"] < 0 or base[i]["srf"] < 0:         # an array's record that appears / stays behind unlinked: the test above
            continue
        a, b = Sp[int(base[i]["srf"])], Se[int(tgt[i]["srf"])]
        words = [w for w in range(38) if a[w] != b[w]]
        assert not words, (i, words)
        assert (int(a.view(np.int32)[39]) < 0) == (int(b.view(np.int32)[39]) < 0), i
    # materials: same table order in both captures (the walker numbers them as it meets them: compare through the surfaces)
    import struct
    hp, he = struct.unpack_from("<4I6I7I5I", patched, 0), struct.unpack_from("<4I6I7I5I", target, 0)
    Mp = np.frombuffer(patched, dtype=np.uint32, count=hp[5] * 32, offset=hp[12]).reshape(-1, 32)
    Me = np.frombuffer(target, dtype=np.uint32, count=he[5] * 32, offset=he[12]).reshape(-1, 32)
    for i in range(len(tgt)):
        if tgt[i]["srf"] < 0 or not (0 <= tgt[i]["tag"] < 9):
            continue
        a, b = Sp[int(base[i]["srf"])].view(np.int32), Se[int(tgt[i]["srf"])].view(np.int32)
        for side in (40, 41):
            if a[side] >= 0:
                ma, mb = Mp[a[side]], Me[b[side]]
                if ma[4] or ma[5]:                          # a one-texel texture ignores its scale: apply leaves it alone
                    assert (ma[:4] == mb[:4]).all(), (i, side, ma[:4].view(np.float32), mb[:4].view(np.float32))
    frame, _, _ = oracle.render(qr.build_lists(patched), threads=4)
    assert np.array_equal(frame, load_frame(target_name) & 0xFFFFFF)


@pytest.mark.gpu
@pytest.mark.parametrize("base_name,target_name", SELF_NODES[::3])
def test_gpu_surfaces_entering_and_leaving_their_own_transform_node(qr, base_name, target_name):
    patched, *_ = _apply_towards(qr, base_name, target_name)
    sc = qr.Scene(qr.build_lists(patched), rebin_tiles=True)
    frame = sc.render().cpu().numpy().view(np.uint32) & 0xFFFFFF
    assert np.array_equal(frame, load_frame(target_name) & 0xFFFFFF)


# Every transform of the scene drawn anew inside the engine (tests/golden/make_tree_golden.py JITTER; frames with the engine's
# tiling off, which is not conservative on such transforms -- tests/test_rebin_pin.py): dozens of surfaces change the array
# that is their transform node at once, arrays under turning arrays start and stop turning, clippers among them.
REGROUPED = [("demo01_160", "demo01_160_jt1"), ("demo02_160", "demo02_160_jt3"), ("demo03_160", "demo03_160_jt3"),
             ("test02_160", "test02_160_jt23"), ("test03_160", "test03_160_jt22"), ("test11_160", "test11_160_jt23"),
             ("test12_160", "test12_160_jt21"), ("test13_160", "test13_160_jt1"), ("test14_160", "test14_160_jt1"),
             ("test16_160", "test16_160_jt3")]


def _regrouped(qr, base_name, target_name):
    tb, base = load_tree(qr, base_name)
    _, tgt = load_tree(qr, target_name)
    nxt = base.copy()
    for f in ("scl", "rot", "pos"):
        nxt[f] = tgt[f]
    st0, st1 = qr.hierarchy_update(base, tb["opts"]), qr.hierarchy_update(nxt, tb["opts"])
    moved = [i for i in range(len(base)) if 0 <= base[i]["tag"] < 9 and int(st0[i]["trnode"]) != int(st1[i]["trnode"])]
    patched = qr.hierarchy_apply(load_blob(base_name), nxt, tb["opts"], camera=tb["camera"], base=base,
                                 flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
    return patched, nxt, st1, moved


@pytest.mark.parametrize("base_name,target_name", REGROUPED)
def test_any_object_may_start_or_stop_turning(qr, oracle, base_name, target_name):
    """qr_hierarchy_apply regroups the global list and every clipper list when surfaces change the array that is their
    transform node (rt_SceneThread::insert / sclip, engine.cpp:1148-1214, 1845-1947), creates the records of new transform
    nodes, and the rebuilt scene renders to the reference's frame of the target, pixel for pixel.  The structure is checked
    against the hierarchy: every array element is followed by exactly the surfaces whose transform node it is."""
    patched, nxt, st1, moved = _regrouped(qr, base_name, target_name)
    assert moved, "the pair does not exercise the case"
    S, E, clist = _snapshot_view(patched)
    node_of = {int(nxt[i]["srf"]): i for i in range(len(nxt)) if nxt[i]["srf"] >= 0 and 0 <= nxt[i]["tag"] < 9}
    e, seen, open_group = clist, [], None
    while e != -1:
        simd, data, nxt_e, kind = (int(x) for x in E[e])
        if data != -1:
            assert open_group is None and int(S[simd].view(np.int32)[37]) == -1          # an array's record, no nesting
            open_group = [simd, data]
        else:
            i = node_of[simd]
            seen.append(i)
            t = int(st1[i]["trnode"])
            if open_group is not None:
                assert t not in (-1, i) and int(S[simd].view(np.int32)[39]) == open_group[0]  # the member's transform node is the head
                if e == open_group[1]:
                    open_group = None
            else:
                assert t in (-1, i)
        e = nxt_e
    assert open_group is None and sorted(seen) == sorted(node_of.values())
    frame, _, _ = oracle.render(qr.build_lists(patched), threads=4)
    assert np.array_equal(frame, tree_frame(target_name) & 0xFFFFFF)


@pytest.mark.gpu
@pytest.mark.parametrize("base_name,target_name", REGROUPED)
def test_gpu_any_object_may_start_or_stop_turning(qr, base_name, target_name):
    patched, *_ = _regrouped(qr, base_name, target_name)
    sc = qr.Scene(qr.build_lists(patched), rebin_tiles=True)
    frame = sc.render().cpu().numpy().view(np.uint32) & 0xFFFFFF
    assert np.array_equal(frame, tree_frame(target_name) & 0xFFFFFF)


def _list_shape(blob, head):
    """A list as (record, kind, x) per element: x = True for an array element / transform-node marker (`ends` tells where its
    run ends, as a position in the list), else the element's data word (a clipper's side, an accum marker's sign)."""
    S, E, _ = _snapshot_view(blob)
    chain, e = [], head
    while e != -1:
        chain.append(e)
        e = int(E[e][2])
    pos = {e: k for k, e in enumerate(chain)}
    out = []
    for e in chain:
        simd, data, nxt, kind = (int(x) for x in E[e])
        if kind == 2 or (simd >= 0 and int(S[simd].view(np.int32)[37]) == -1):
            out.append((simd, kind, True, pos[data]))
        else:
            out.append((simd, kind, data, None))
    return out


def _all_tree_cases():
    out = []
    for f in sorted(os.listdir(TREE)):
        if f.endswith(".json.gz"):
            name = f[:-len(".json.gz")]
            if name in MANIFEST or os.path.exists(os.path.join(TREE, name + ".qrs.gz")):
                out.append(name)
    return out


@pytest.mark.parametrize("name", _all_tree_cases())
def test_regrouping_an_unchanged_scene_gives_the_engines_own_lists(qr, name):
    """QR_HIER_REGROUP on the engine's capture with its own node table: the global list and EVERY clipper list are rebuilt from
    their leaves by the code that serves a changing set of transform nodes -- and come out with the ENGINE's structure, element
    for element: array elements and transform-node markers in front of the same members, accum markers where they were
    (rt_SceneThread::insert / sclip, engine.cpp:1148-1214, 1845-1947).  Every captured tree: test and demo scenes, jittered,
    moved, swarms."""
    t, nodes = load_tree(qr, name)
    blob = load_blob(name) if name in MANIFEST else _fuzz_blob(name)
    out = qr.hierarchy_apply(blob, nodes, t["opts"], camera=t["camera"], base=nodes,
                             flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS | qr.HIER_REGROUP)
    S0, E0, c0 = _snapshot_view(blob)
    S1, E1, c1 = _snapshot_view(out)
    assert len(S1) == len(S0)                                           # no transform node without a record
    assert c1 >= len(E0) or c0 == -1                                    # ... and the list IS rebuilt (fresh elements)
    assert _list_shape(out, c1) == _list_shape(blob, c0)
    n_lists = 0
    for r in range(len(S0)):
        h0, h1 = int(S0[r].view(np.int32)[38]), int(S1[r].view(np.int32)[38])
        assert (h0 < 0) == (h1 < 0)
        if h0 >= 0:
            assert h1 >= len(E0)
            assert _list_shape(out, h1) == _list_shape(blob, h0), r
            n_lists += 1
    assert (S1[:, :38] == S0[:, :38]).all()


def test_turning_an_array_and_turning_it_back_restores_the_snapshot(qr, oracle):
    """Demo scene 2's frame table (nested arrays with bounding volumes, clipped legs; obj_frametable.h) turned by 10 degrees:
    its surfaces leave the transform node above for a new one, whose record is n_srf + 0 (node order); patching the PATCHED
    snapshot with the original transforms regroups everything back: every record field, the global list and every clipper
    list are the original's, and the frame is the reference's."""
    t2, b2 = load_tree(qr, "demo02_160")
    arr = next(i for i in range(len(b2)) if b2[i]["tag"] == -1 and b2[i]["bvb"] >= 0 and b2[i]["parent"] >= 0)
    turned = b2.copy()
    turned[arr]["rot"][2] += 10.0
    flags = qr.HIER_RESET_TILES | qr.HIER_BOUNDS
    original = load_blob("demo02_160")
    there = qr.hierarchy_apply(original, turned, t2["opts"], camera=t2["camera"], base=b2, flags=flags)
    S0, _, c0 = _snapshot_view(original)
    S1, _, c1 = _snapshot_view(there)
    assert len(S1) > len(S0)                                            # a new transform node's record
    turned_after = qr.hierarchy_records_after_apply(original, turned, t2["opts"])
    new = [i for i in range(len(b2)) if turned_after[i]["srf"] != b2[i]["srf"]]
    assert new and all(b2[i]["tag"] == -1 and turned_after[i]["srf"] >= len(S0) for i in new)
    assert [x[0] for x in _list_shape(there, c1) if x[2] is True and x[0] >= len(S0)] != []    # ... heads a group of the list
    frame_there, _, _ = oracle.render(qr.build_lists(there), threads=4)
    assert not np.array_equal(frame_there, load_frame("demo02_160") & 0xFFFFFF)
    back_nodes = turned_after.copy()
    for f in ("scl", "rot", "pos"):
        back_nodes[f] = b2[f]
    back = qr.hierarchy_apply(there, back_nodes, t2["opts"], camera=t2["camera"], base=turned_after, flags=flags)
    S2, _, c2 = _snapshot_view(back)
    for r in range(len(S0)):
        live = int(S0[r].view(np.int32)[37]) >= 0 or any(int(b2[i]["srf"]) == r and True for i in range(len(b2)))
        words = [w for w in range(38) if S0[r][w] != S2[r][w]]
        assert not live or not words, (r, words)
    assert _list_shape(back, c2) == _list_shape(original, c0)
    heads0 = sorted(set(int(S0[r].view(np.int32)[38]) for r in range(len(S0))))
    for r in range(len(S0)):
        h0, h2 = int(S0[r].view(np.int32)[38]), int(S2[r].view(np.int32)[38])
        assert (h0 < 0) == (h2 < 0)
        if h0 >= 0:
            assert _list_shape(back, h2) == _list_shape(original, h0), r
    frame, _, _ = oracle.render(qr.build_lists(back), threads=4)
    assert np.array_equal(frame, load_frame("demo02_160") & 0xFFFFFF)


def test_surfaces_the_engine_removed_from_its_camera_list_come_back(qr, oracle):
    """rt_SceneThread::insert drops a surface that is fully hidden behind another from the camera list (the 4|x results of
    bbox_sort, rtgeom.cpp:1551-1560; engine.cpp:1293-1330): such a list is good for its own frame only.  A snapshot whose list
    lacks a surface of the table (here: unlinked by hand) gets it back when it is patched with `base` and the tiles reset."""
    import struct
    t, base = load_tree(qr, "test05_160")
    blob = bytearray(load_blob("test05_160"))
    S, E, clist = _snapshot_view(bytes(blob))
    h = struct.unpack_from("<4I6I7I5I", blob, 0)
    chain, e = [], clist
    while e != -1:
        chain.append(e)
        e = int(E[e][2])
    assert len(chain) >= 2 and all(int(E[e][1]) == -1 for e in chain)   # plain surfaces: cut the last one (the floor) off
    struct.pack_into("<i", blob, h[14] + 16 * chain[-2] + 8, -1)
    gone = int(E[chain[-1]][0])
    for k in range(h[8]):                                               # ... and every tile gets that shortened list
        struct.pack_into("<i", blob, h[15] + 4 * k, clist)
    lame, _, _ = oracle.render(bytes(blob), threads=4)
    assert not np.array_equal(lame, load_frame("test05_160") & 0xFFFFFF)
    patched = qr.hierarchy_apply(bytes(blob), base, t["opts"], camera=t["camera"], base=base, flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
    assert gone in [x[0] for x in _list_shape(patched, _snapshot_view(patched)[2])]
    frame, _, _ = oracle.render(qr.build_lists(patched), threads=4)
    assert np.array_equal(frame, load_frame("test05_160") & 0xFFFFFF)


def test_own_transform_node_changes_outside_the_scope_are_refused(qr):
    """Without the materials' own texture scale in the node table (qr_node.tex) a scaled, textured plane's cannot be read
    back from the snapshot: refused, not guessed."""
    with pytest.raises(qr.QrError, match="scaled plane"):
        _apply_towards(qr, "test12_160_j8", "test12_160_noopt", with_tex=False)


def test_updates_outside_the_scope_are_refused(qr):
    """What apply still refuses.  A changing set of transform nodes needs the node tables of both times (`base`) and the tile
    lists reset."""
    t0, base = load_tree(qr, "demo01_160")
    _, tgt = load_tree(qr, "demo01_160_t2500")
    nxt = base.copy()
    for f in ("scl", "rot", "pos"):
        nxt[f] = tgt[f]
    with pytest.raises(qr.QrError, match="transform node"):
        qr.hierarchy_apply(load_blob("demo01_160"), nxt, t0["opts"], camera=t0["camera"])               # no base table
    with pytest.raises(qr.QrError, match="QR_HIER_RESET_TILES"):
        qr.hierarchy_apply(load_blob("demo01_160"), nxt, t0["opts"], camera=t0["camera"], base=base)    # tile lists would go stale
    # a right-angle turn of a clipped surface changes its axis mapping: the clip box would have to be rebuilt
    tb, b13 = load_tree(qr, "test13_160")
    k = next(i for i in range(len(b13)) if 0 <= b13[i]["tag"] < 9 and b13[i]["srf"] >= 0)
    turned = b13.copy()
    turned[k]["rot"][0] += 90.0
    with pytest.raises(qr.QrError, match="axis mapping|transform node"):
        qr.hierarchy_apply(load_blob("test13_160"), turned, tb["opts"], base=b13)
    # malformed tables
    bad = b13.copy()
    bad[1]["parent"] = 5
    with pytest.raises(qr.QrError, match="precede"):
        qr.hierarchy_update(bad, tb["opts"])
    bad = b13.copy()
    bad[k]["srf"] = 100000
    with pytest.raises(qr.QrError, match="out of range"):
        qr.hierarchy_apply(load_blob("test13_160"), bad, tb["opts"])


@pytest.mark.parametrize("name", ["demo01_160_aa2_t2500", "test13_160", "test16_160_noopt"])
def test_tree_dump_through_the_engine_reproduces_the_fixture(name, tmp_path):
    """The fixture generator's path -- the unmodified engine renders, the shim captures the snapshot, qr_capture_index maps
    the engine's records to snapshot indices, ref_driver dumps the tree -- gives the committed tree again (needs
    oracle/_ref, which exists in the build container and travels to the GPU box)."""
    import subprocess
    from conftest import ROOT
    shim = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
    if not os.path.exists(shim):
        pytest.skip("oracle/_ref/qr_ref_shim is not built")
    m = MANIFEST[name]
    (tmp_path / "dump").mkdir()
    out, snap = tmp_path / "t.json", tmp_path / "s.qrs"
    cmd = [shim, "--scene", m["scene"], "-w", str(m["w"]), "-h", str(m["h"]), "--snapshot", str(snap), "--tree", str(out)] + m["args"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True, timeout=120)
    with open(os.path.join(TREE, name + ".json.gz"), "rb") as f:
        want = json.loads(gzip.decompress(f.read()))
    assert json.loads(out.read_text()) == want
    assert snap.read_bytes() == load_blob(name)
