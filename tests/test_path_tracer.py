"""Path-tracer mode (the reference's RT_FEAT_PT).

Two modes, both against frames of the reference itself (tests/golden/pt/, made by tests/golden/make_pt_golden.py from
oracle/_ref with the smallpt Cornell box, test18; the oracle has no path tracer and is not involved):

* EAGER (set_pt(True, eager=True)): PINNED, bit-exact.  The reference shades eagerly -- every hit that passes the depth
  test while a list is walked is shaded at once, and in path-tracer mode that shading draws random numbers -- so a
  sample's stream depends on that order.  The eager machine (csrc/qr_pt_eager.hpp) follows it and reproduces the
  reference's frames pixel for pixel: recursion depths 0..10, 1 to 512 accumulated frames, 4x FSAA, Gamma, Fresnel.
* STATISTICAL (set_pt(True)): the fast kernel with deferred shading: same generator, per-sample seeds
  (rt_Scene::reset_pseed), sampling formulas (tent-filter jitter, cosine hemisphere with the reference's power series,
  Russian roulette, Fresnel split) and running-mean accumulation, numbers drawn for the final hit of a walk only.  Its
  frames have the reference's DISTRIBUTION: statistics of N accumulated frames are compared; its first frame at
  recursion depth 0, where no shading order exists, is the reference's pixel for pixel.
"""
import gzip
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PT = os.path.join(ROOT, "tests", "golden", "pt")


def _blob(name):
    return gzip.decompress(open(os.path.join(PT, name + ".qrs.gz"), "rb").read())


def _ref(name, w, h):
    return np.frombuffer(gzip.decompress(open(os.path.join(PT, name + ".raw.gz"), "rb").read()), dtype=np.uint32).reshape(h, w)


def _rgb(a):
    return np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.float64)


def _blocks(a, b=8):
    h, w, _ = a.shape
    return a.reshape(h // b, b, w // b, b, 3).mean((1, 3))


def test_pt_snapshot_carries_the_flag_and_the_emission():
    """The walker captures path-tracer snapshots (qr_frame.pt_on, qr_material.emis = mat_COL_R/G/B); ordinary snapshots
    keep zeros there, so every older fixture is unchanged."""
    b = _blob("test18_160_pt")
    hdr = struct.unpack_from("<32I", b, 0)
    n_mat, off_frame, off_mat = hdr[5], hdr[10], hdr[12]
    fr = np.frombuffer(b, dtype=np.int32, count=49, offset=off_frame)
    assert fr[41] == 1                                        # qr_frame.pt_on
    m = np.frombuffer(b, dtype=np.float32, count=n_mat * 32, offset=off_mat).reshape(n_mat, 32)
    emis = m[:, 21:24]
    assert (emis == 12.0).all(axis=1).sum() >= 1              # smallpt's light: emission (12, 12, 12)
    assert (emis == 0.0).all(axis=1).sum() >= n_mat - 2
    plain = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "test18_160.qrs.gz"), "rb").read())
    h2 = struct.unpack_from("<32I", plain, 0)
    m2 = np.frombuffer(plain, dtype=np.float32, count=h2[5] * 32, offset=h2[12]).reshape(h2[5], 32)
    assert (m2[:, 21:24] == 0.0).all()
    assert np.frombuffer(plain, dtype=np.int32, count=49, offset=h2[10])[41] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("snap,ref,n,mean_tol,block_tol", [
    ("test18_160_pt", "test18_160_pt_n64", 64, 1.0, 8.0),
    ("test18_160_pt", "test18_160_pt_n512", 512, 0.6, 3.0),
    ("test18_160_gf_aa4_pt", "test18_160_gf_aa4_pt_n128", 128, 0.8, 3.0),
])
def test_gpu_path_tracer_has_the_references_distribution(qr, snap, ref, n, mean_tol, block_tol):
    """N accumulated frames against N accumulated frames of the reference: channel means of the whole frame and the
    means of 8x8 pixel blocks.  Measured on MI355X: channel means within 0.2 of 255, block means 5.1 / 1.8 / 1.7 apart
    on average (two independent estimates of the same image: the distance shrinks with sqrt(N))."""
    import torch
    scn = qr.Scene(_blob(snap))
    want = _rgb(_ref(ref, scn.width, scn.height))
    plain = _rgb(scn.render().cpu().numpy().view(np.uint32))         # ray-traced frame of the same snapshot
    scn.set_pt(True)
    f = scn.new_frame()
    for _ in range(n):
        scn.render(f)
    torch.cuda.synchronize()
    got = _rgb(f.cpu().numpy().view(np.uint32))
    assert np.abs(got.mean((0, 1)) - want.mean((0, 1))).max() < mean_tol
    assert np.abs(_blocks(got) - _blocks(want)).mean() < block_tol
    # the statistic can tell images apart: the ray-traced frame of the same scene is far away from the reference's
    assert np.abs(plain.mean((0, 1)) - want.mean((0, 1))).max() > 20.0
    assert np.abs(_blocks(plain) - _blocks(want)).mean() > 4 * block_tol


@pytest.mark.gpu
def test_gpu_path_tracer_is_deterministic_and_restartable(qr, oracle):
    """Same seeds, same frames: two accumulations of 16 frames are bit-identical; set_pt(True) restarts the mean;
    set_pt(False) gives the ray-traced frame again, which is the oracle's (emission and the flag change nothing there)."""
    import torch
    blob = _blob("test18_160_pt")
    scn = qr.Scene(blob)
    runs = []
    for _ in range(2):
        scn.set_pt(True)
        f = scn.new_frame()
        for _ in range(16):
            scn.render(f)
        torch.cuda.synchronize()
        runs.append(f.cpu().numpy().copy())
    assert (runs[0] == runs[1]).all()
    one = scn.new_frame()
    scn.set_pt(True); scn.render(one); torch.cuda.synchronize()
    assert (one.cpu().numpy() != runs[0]).any()                       # one sample is not the mean of sixteen
    scn.set_pt(False)
    rt = scn.render(); torch.cuda.synchronize()
    o_frame, _, _ = oracle.render(blob, threads=4)
    assert (rt.cpu().numpy().view(np.uint32) == o_frame).all()
    scn.set_pt(True)
    with pytest.raises(qr.QrError):
        scn.render_count()                                            # counting renders are refused in this mode
    with pytest.raises(qr.QrError):
        qr.MultiRender([(scn, scn.new_frame(), 0, scn.height)])()    # and so are multi-target launches
    # a path-traced frame is one more sample of EVERY pixel: cut into row-range launches the sample count would advance
    # once per launch and weigh the later ranges wrongly -- refused, and nothing is counted by the refused call
    two = scn.new_frame(); scn.render(two); torch.cuda.synchronize()
    for sel in ((scn.set_rows, (0, scn.height // 2)), (scn.set_rows, (0, scn.height, 1, 2)), (scn.set_tile_rows, (1, 2))):
        sel[0](*sel[1])
        with pytest.raises(qr.QrError):
            scn.render(two)
        with pytest.raises(qr.QrError):
            scn.render_host()
    scn.set_rows(0, scn.height)
    scn.set_pt(True)
    again = scn.new_frame(); scn.render(again); torch.cuda.synchronize()
    assert (again.cpu().numpy() == one.cpu().numpy()).all()           # whole frames work as before


@pytest.mark.gpu
@pytest.mark.parametrize("ref,fsaa", [("test18_160_pt_d0_n1", 0), ("test18_160_aa4_pt_d0_n1", 2)])
def test_gpu_path_tracer_first_frame_depth0_is_bit_exact(qr, ref, fsaa):
    """qr_ref --scene test18 --depth 0 --pt 1 [--fsaa 4]: the reference's first path-traced frame without recursion."""
    import torch
    b = bytearray(_blob("test18_160_pt"))
    if fsaa:
        # same scene with 4x FSAA: the engine's sample offsets (engine.cpp:3525-3546 pattern, as in the aa4 snapshot)
        a = _blob("test18_160_gf_aa4_pt")
        oa, ob = struct.unpack_from("<I", a, 40)[0], struct.unpack_from("<I", b, 40)[0]
        fa = np.frombuffer(a, dtype=np.uint32, count=49, offset=oa)
        fb = np.frombuffer(bytes(b), dtype=np.uint32, count=49, offset=ob).copy()
        fb[10:18] = fa[10:18]; fb[30] = fa[30]                 # hor_a / ver_a, fsaa
        b[ob:ob + 196] = fb.tobytes()
    scn = qr.Scene(bytes(b))
    scn.set_depth(0)
    scn.set_pt(True)
    f = scn.render(); torch.cuda.synchronize()
    want = _ref(ref, scn.width, scn.height)
    assert int((want != 0).sum()) > 100
    assert int((f.cpu().numpy().view(np.uint32) != want).sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("depth,frames,allowed", [(2, 2, 0), (6, 2, 0), (8, 1, 0), (8, 2, 0), (10, 1, 0), (10, 2, 0), (10, 4, 0)])
def test_gpu_eager_path_tracer_reproduces_the_references_frames(qr, depth, frames, allowed):
    """set_pt(True, eager=True): shading in the reference's order (csrc/qr_pt_eager.hpp) -- every hit that passes the
    depth test is shaded at once, bounce subtree first, then the Fresnel split, refraction, reflection -- so a sample
    consumes the reference's numbers.  Pixel for pixel the reference's frames (qr_ref --pt N) at every recursion depth up
    to the default 10, first frame and accumulated ones (frame k's jitter depends on every number frames 1..k-1 drew)."""
    import torch
    scn = qr.Scene(_blob("test18_160_pt"))
    scn.set_depth(depth)
    scn.set_pt(True, eager=True)
    f = scn.new_frame()
    for _ in range(frames):
        scn.render(f)
    torch.cuda.synchronize()
    want = _ref("test18_160_pt_d%d_n%d" % (depth, frames), scn.width, scn.height)
    assert int((f.cpu().numpy().view(np.uint32) != want).sum()) <= allowed


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [64, 512])
def test_gpu_eager_path_tracer_accumulates_the_references_image(qr, frames):
    """N accumulated frames at the default depth: every one of the N x 19 200 streams stays the reference's to the end."""
    import torch
    scn = qr.Scene(_blob("test18_160_pt"))
    scn.set_pt(True, eager=True)
    f = scn.new_frame()
    for _ in range(frames):
        scn.render(f)
    torch.cuda.synchronize()
    want = _ref("test18_160_pt_n%d" % frames, scn.width, scn.height)
    assert int((f.cpu().numpy().view(np.uint32) != want).sum()) == 0


@pytest.mark.gpu
def test_gpu_eager_path_tracer_with_fsaa_gamma_fresnel(qr):
    """4x FSAA (four samples per pixel, each with its own stream), Gamma, Fresnel (the split of Fresnel surfaces between
    reflection and refraction draws numbers): the reference's first three accumulated frames, pixel for pixel."""
    import torch
    scn = qr.Scene(_blob("test18_160_gf_aa4_pt"))
    scn.set_pt(True, eager=True)
    f = scn.new_frame()
    for _ in range(3):
        scn.render(f)
    torch.cuda.synchronize()
    want = _ref("test18_160_gf_aa4_pt_n3", scn.width, scn.height)
    assert int((want != 0).sum()) > 1000
    assert int((f.cpu().numpy().view(np.uint32) != want).sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test18_160", "demo01_160", "demo02_160", "demo03_160", "test09_160", "test13_160", "test16_160",
                                  "swarm_demo01_240_mix", "test11_160_j7"])
def test_gpu_eager_machine_with_ray_tracer_shading_is_the_ray_tracer(qr, name):
    """Self-test of the eager machine: with the ray tracer's shading (lights, shadows, no random numbers) shading every
    provisional hit cannot change a pixel, so its frame is the deferred kernel's, at every depth -- also on the scenes
    whose surfaces carry clipper programs (CSG), which the machine's walk runs per lane."""
    import torch
    blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", name + ".qrs.gz"), "rb").read())
    scn = qr.Scene(blob)
    for depth in (10, 3, 0):
        scn.set_depth(depth)
        rt = scn.render(); torch.cuda.synchronize()
        qr._check(qr.lib().qr_scene_set_pt(scn._h, 3))
        f = scn.render(); torch.cuda.synchronize()
        qr._check(qr.lib().qr_scene_set_pt(scn._h, 0))
        assert int((f != rt).sum().item()) == 0
