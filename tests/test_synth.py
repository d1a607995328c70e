"""Synthetic "10k quadrics" scene (BASELINE.json config 5, quadray-engine_amd/synth.py).

The reference cannot produce this scene (SURVEY.md 8(c) limit 3), so its parity chain is:
oracle pinned bit-exactly by the reference on demo01-03 / test01-18  ->  oracle renders the synthetic
snapshot  ->  HIP backend must equal the oracle.  "Parity unpinned" with respect to the reference
itself for these inputs; the shapes, materials and list format are the ones the pinned cases use.
"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _synth():
    spec = importlib.util.spec_from_file_location("qr_synth", os.path.join(ROOT, "quadray-engine_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


SMALL = dict(n_objects=300, width=320, height=240, depth=4, box=20.0)
SMALL_HASH = 0xd42a6be53b20d038
MID = dict(n_objects=2000, width=384, height=216, depth=4, box=45.0, gamma=True, fsaa=2)


def test_synth_scene_is_deterministic_and_bounding_volumes_are_conservative(oracle):
    synth = _synth()
    blob = synth.make_scene(**SMALL)
    assert blob == synth.make_scene(**SMALL)
    frame, ids, counts = oracle.render(blob, threads=8, want_ids=True)
    assert oracle.frame_hash(frame) == SMALL_HASH
    assert counts["reflect"] > 0 and counts["refract"] > 0 and counts["shadow"] > counts["primary"]
    assert len(np.unique(ids)) > 100                       # most objects are somebody's visible hit
    # the two-level array hierarchy must not change a pixel against the flat list
    flat, flat_ids, _ = oracle.render(synth.make_scene(hierarchy=False, **SMALL), threads=8, want_ids=True)
    assert (flat == frame).all()
    # deferred shading (what the HIP backend does) gives the reference semantics' pixels
    deferred, _, _ = oracle.render(blob, threads=8, deferred=True)
    assert (deferred == frame).all()


@pytest.mark.gpu
@pytest.mark.parametrize("lists", ["generator", "built"])
@pytest.mark.parametrize("cfg", [SMALL, MID], ids=["300obj_320x240", "2000obj_384x216_aa4_gamma"])
def test_gpu_synth_scene_matches_oracle(qr, oracle, cfg, lists):
    """No tile lists in the snapshot: the backend bins them on the GPU, then must match the oracle.  Per-object shadow
    lists either from the generator or built by the product's pass from the global list (what bench.py uses)."""
    import torch
    blob = _synth().make_scene(**cfg) if lists == "generator" else qr.build_lists(_synth().make_scene(shadow_lists=False, **cfg))
    scn = qr.Scene(blob, rebin_tiles=True)
    assert scn.info.n_tiles > 1
    frame = scn.new_frame(); ids = torch.full_like(frame, -2)
    scn.render(frame, ids=ids); torch.cuda.synchronize()
    o_frame, o_ids, _ = oracle.render(blob, threads=16, want_ids=True)
    out = frame.cpu().numpy().view(np.uint32)
    assert int((out != o_frame).sum()) == 0
    assert (ids.cpu().numpy() == o_ids).all()
    _, _, o_counts = oracle.render(blob, threads=16, deferred=True)
    _, c = scn.render_count()
    assert c.as_dict() == {k: o_counts[k] for k in c.as_dict()}
    # without the binning pass the single whole-frame tile is still correct (only slower)
    plain = qr.Scene(blob)
    f2 = plain.render(); torch.cuda.synchronize()
    assert bool((f2 == frame).all())


@pytest.mark.gpu
def test_gpu_synth_10k_full_size_properties(qr):
    """BASELINE.json config 5 at full size (10 000 quadrics, 7680x4320, depth 4): too large for the CPU
    oracle, so size-independent properties: tile-row shards compose to the whole frame bit-exactly, two
    tile sizes of the binning pass agree, and a 1/16-scale render equals the oracle-checked path's."""
    import torch
    blob = qr.build_lists(_synth().make_scene(shadow_lists=False))      # config 5 as bench.py renders it
    scn = qr.Scene(blob, rebin_tiles=True)
    whole = scn.render(); torch.cuda.synchronize()
    assert int((whole != 0).sum().item()) > whole.numel() // 4
    acc = torch.zeros_like(whole)
    for r in range(3):
        scn.set_tile_rows(r, 3)
        scn.render(acc)
    torch.cuda.synchronize()
    assert bool((acc == whole).all())
    os.environ["QR_BIN_TILE"] = "16x16"
    try:
        other = qr.Scene(blob, rebin_tiles=True)
    finally:
        del os.environ["QR_BIN_TILE"]
    assert (other.info.tile_w, other.info.tile_h) == (16, 16)
    f2 = other.render(); torch.cuda.synchronize()
    assert bool((f2 == whole).all())


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,fsaa", [(1, 1, 0), (7, 5, 0), (33, 17, 2), (100, 3, 0), (9, 64, 2), (257, 31, 0)])
def test_gpu_ragged_frames_match_oracle(qr, oracle, w, h, fsaa):
    """Frame sizes that are not multiples of the wave footprint (8x8, 4x4 with FSAA) or of the tile size,
    down to a single pixel; with and without the GPU binning pass."""
    import torch
    blob = _synth().make_scene(n_objects=40, width=w, height=h, depth=3, box=8.0, fsaa=fsaa, gamma=bool(fsaa))
    o_frame, _, _ = oracle.render(blob, threads=4)
    for rebin in (False, True):
        scn = qr.Scene(blob, rebin_tiles=rebin)
        f = scn.render(); torch.cuda.synchronize()
        assert (f.cpu().numpy().view(np.uint32) == o_frame).all(), f"rebin={rebin}"


@pytest.mark.gpu
def test_gpu_scene_without_objects(qr, oracle):
    """Only the ground plane: every list is one cell long, most tiles are empty."""
    import torch
    blob = _synth().make_scene(n_objects=0, width=64, height=48, depth=4, box=10.0)
    o_frame, _, counts = oracle.render(blob, threads=2)
    assert counts["reflect"] == 0 and counts["refract"] == 0
    for rebin in (False, True):
        f = qr.Scene(blob, rebin_tiles=rebin).render(); torch.cuda.synchronize()
        assert (f.cpu().numpy().view(np.uint32) == o_frame).all()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"QR_CULL": "0"}, {"QR_CULL": "1"}, {"QR_BIN_TILE": "8x8"},
                                 {"QR_GRID": "0"}, {"QR_GRID": "64", "QR_CULL": "0"},
                                 {"QR_DDA": "0"}, {"QR_DDA": "64", "QR_DDA_CELLS": "0.2"}, {"QR_DDA_CELLS": "16"}])
def test_gpu_synth_build_variants_match_oracle(qr, oracle, env):
    """Upload-time variants of the compiled scene: no bounding-sphere cull cells, cull on planes only, 8x8 tiles
    from the binning pass, without / with a lower threshold for the shadow lists by hit position, without the uniform
    grids of long lists (secondary rays walk the hierarchy with hand-over then) and with very coarse / very fine ones:
    same pixels, hit ids and ray counts as the oracle."""
    import torch
    blob = _synth().make_scene(**MID)
    if "QR_GRID" in env or "QR_DDA" in env or "QR_DDA_CELLS" in env:
        blob = qr.build_lists(_synth().make_scene(shadow_lists=False, **MID))     # own light list per surface: grids apply
    os.environ.update(env)
    try:
        scn = qr.Scene(blob, rebin_tiles=True)
    finally:
        for k in env:
            del os.environ[k]
    frame = scn.new_frame(); ids = torch.full_like(frame, -2)
    scn.render(frame, ids=ids); torch.cuda.synchronize()
    o_frame, o_ids, _ = oracle.render(blob, threads=16, want_ids=True)
    assert (frame.cpu().numpy().view(np.uint32) == o_frame).all()
    assert (ids.cpu().numpy() == o_ids).all()
    _, _, o_counts = oracle.render(blob, threads=16, deferred=True)
    _, c = scn.render_count()
    assert c.as_dict() == {k: o_counts[k] for k in c.as_dict()}


def _recamera(blob, seed, scale):
    """The snapshot's camera moved and turned at random (same rotation for view, horizontal and vertical vectors)."""
    import struct
    rng = np.random.default_rng(seed)
    b = bytearray(blob)
    off_frame = struct.unpack_from("<I", b, 4 * 10)[0]
    fr = np.frombuffer(b, dtype=np.float32, count=49, offset=off_frame).copy()
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = rng.uniform(0.0, 0.6)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
    for o in (1, 4, 7):
        fr[o:o + 3] = R @ fr[o:o + 3]
    fr[25:28] += rng.uniform(-scale, scale, size=3).astype(np.float32)
    b[off_frame:off_frame + 196] = fr.tobytes()
    return bytes(b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gpu_synth_10k_objects_match_oracle(qr, oracle, seed):
    """All 10 000 objects of BASELINE.json config 5 at frame sizes the oracle finishes in seconds (1920x1080 / 640x360, depth 4),
    lists from the product's list-building pass as in bench.py: the per-lane walk with work hand-over (walk_pool), the
    shadow lists by hit position on the ground plane (CGrid) and the binning pass against the oracle, which knows
    none of them -- the original camera and two random ones (seed > 0: also from inside the object cloud)."""
    import torch
    w, h = (1920, 1080) if seed == 0 else (640, 360)
    blob = qr.build_lists(_synth().make_scene(shadow_lists=False, width=w, height=h))
    if seed:
        blob = _recamera(blob, 77 * seed, 30.0)
    assert qr.program_stats(blob).n_grids == 4
    scn = qr.Scene(blob, rebin_tiles=True)
    frame = scn.new_frame(); ids = torch.full_like(frame, -2)
    scn.render(frame, ids=ids); torch.cuda.synchronize()
    o_frame, o_ids, _ = oracle.render(blob, threads=16, want_ids=True)
    out = frame.cpu().numpy().view(np.uint32)
    assert int((out != o_frame).sum()) == 0
    assert (ids.cpu().numpy() == o_ids).all()
    _, _, o_counts = oracle.render(blob, threads=16, deferred=True)
    _, c = scn.render_count()
    assert c.as_dict() == {k: o_counts[k] for k in c.as_dict()}


@pytest.mark.gpu
def test_gpu_synth_10k_full_size_bands_match_oracle(qr, oracle):
    """BASELINE config 5 at its FULL size (7680x4320, depth 4, 10 000 objects, lists and tile lists as bench.py builds them):
    the oracle cannot walk the whole frame in test time, so three bands of rows -- top of the object cloud, the middle, the
    ground plane near the bottom: 144 rows, 1.1 M pixels -- are compared pixel for pixel, and the frame's ray counts must be
    those of the counting kernel variant twice in a row (determinism at 33 M pixels).  Until round 4 this check only ran inside
    `bench.py --workload synth10k_4320p`."""
    import torch
    blob = qr.build_lists(_synth().make_scene(shadow_lists=False, width=7680, height=4320, depth=4))
    scn = qr.Scene(blob, rebin_tiles=True)
    assert (scn.width, scn.height) == (7680, 4320)
    frame = scn.render(); torch.cuda.synchronize()
    out = frame.cpu().numpy().view(np.uint32)
    for r0 in (1200, 2136, 3600):
        band, _, _ = oracle.render(blob, threads=16, rows=(r0, r0 + 48))
        assert int((out[r0:r0 + 48] != band[r0:r0 + 48]).sum()) == 0, f"rows {r0}..{r0 + 48} differ from the oracle"
    _, c1 = scn.render_count(); _, c2 = scn.render_count()
    assert c1.as_dict() == c2.as_dict() and c1.primary == 7680 * 4320
    again = scn.render(); torch.cuda.synchronize()
    assert bool((again == frame).all())


@pytest.mark.gpu
def test_gpu_synth_multi_target_launch_matches_whole_frame(qr):
    """The multi-target launch of the kernel instance with the per-lane walks (qr_render_multi_kernel<.., true>): three
    row blocks of the 2000-object scene (walk_pool, shadow grids) and a block of an engine scene in ONE launch equal
    the whole-frame renders -- what a rank of a sharded step does with a scene of this kind."""
    import torch
    import gzip
    a = qr.Scene(qr.build_lists(_synth().make_scene(shadow_lists=False, **MID)), rebin_tiles=True)
    with open(os.path.join(ROOT, "tests", "golden", "demo01_160.qrs.gz"), "rb") as f:
        b = qr.Scene(gzip.decompress(f.read()))
    whole_a = a.render(); whole_b = b.render(); torch.cuda.synchronize()
    ha, hb = a.height, b.height
    cuts = [0, 56, 64, ha]
    fa = [torch.full_like(whole_a, 0x55) for _ in range(3)]
    fb = torch.full_like(whole_b, 0x55)
    qr.MultiRender([(a, fa[i], cuts[i], cuts[i + 1]) for i in range(3)] + [(b, fb, 8, hb - 8)])()
    torch.cuda.synchronize()
    for i in range(3):
        assert bool((fa[i][cuts[i]:cuts[i + 1]] == whole_a[cuts[i]:cuts[i + 1]]).all())
        assert bool((torch.cat([fa[i][:cuts[i]], fa[i][cuts[i + 1]:]]) == 0x55).all())
    assert bool((fb[8:hb - 8] == whole_b[8:hb - 8]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,box", [(1, 700, 14.0), (2, 1500, 22.0), (3, 3000, 60.0), (4, 900, 9.0)])
def test_gpu_random_synth_scenes_match_oracle(qr, oracle, seed, n, box):
    """Other object clouds than BASELINE's: from very dense (overlapping objects: rays inside glass bodies, many equal
    or near-equal depths where list order decides) to sparse, more recursion, with low thresholds so that every long
    list gets a uniform grid and the ground plane its shadow grids: pixels, hit ids and ray counts as the oracle."""
    import torch
    kw = dict(n_objects=n, width=320, height=180, depth=6, box=box, seed=seed)
    blob = qr.build_lists(_synth().make_scene(shadow_lists=False, **kw))
    os.environ.update({"QR_DDA": "64", "QR_GRID": "64"})
    try:
        assert qr.program_stats(blob).n_dda >= 1
        scn = qr.Scene(blob, rebin_tiles=True)
    finally:
        del os.environ["QR_DDA"], os.environ["QR_GRID"]
    frame = scn.new_frame(); ids = torch.full_like(frame, -2)
    scn.render(frame, ids=ids); torch.cuda.synchronize()
    o_frame, o_ids, _ = oracle.render(blob, threads=16, want_ids=True)
    assert int((frame.cpu().numpy().view(np.uint32) != o_frame).sum()) == 0
    assert (ids.cpu().numpy() == o_ids).all()
    _, _, o_counts = oracle.render(blob, threads=16, deferred=True)
    _, c = scn.render_count()
    assert c.as_dict() == {k: o_counts[k] for k in c.as_dict()}


@pytest.mark.gpu
def test_gpu_guarded_build_finds_no_bad_cell_offset():
    """The QR_STATS + QR_GUARD build of the kernels (every cell offset of the per-lane walks -- walk_div, walk_pool with its
    hand-over, walk_dda -- checked before it is loaded) renders synthetic crowds of 300 / 490 (dense, lowered grid thresholds: a 300 MB image) / 2 000 / 10 000 objects and a swarm
    fixture: no bad offset, frames equal to the oracle's.  (Round 2's memory-access fault was such an offset)"""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_guard.so")
    if not os.path.exists(lib):
        pytest.skip("libqrhip_guard.so not built (make -C quadray-engine_amd/csrc guard)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_guard_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("frame_ok 1") == 5 and "QR_GUARD" not in out.stdout, out.stdout
