"""Several devices and caller-owned frames behind the C ABI (include/qrhip.h: QR_DEVICES, qr_frame_register).

QR_DEVICES=0,1,... cuts the frame of a host-frame call into bands of tile rows, one per entry; every entry renders its band
on its device and copies it into the host frame (the engine's own threads share a frame the same way: rows index, index +
thnum, ... written in place, tracer.cpp:1144-1145, engine.cpp:3465-3478).  A one-GPU box lists its ordinal several times:
separate buffers and streams per entry, the same code path.  The frames must be the single-device frames bit for bit."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import MANIFEST, ROOT, load_blob, load_frame

CASES = ["demo01_160", "demo02_160_gf_aa4", "demo03_160_aa2_t2500", "swarm_demo01_240_mix"]
SHIM = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")


def test_frame_registration_needs_a_device(qr):
    """no GPU: the registration fails loudly like every device entry point (no CPU fallback); with one: round trip."""
    arr = np.zeros((64, 64), dtype=np.uint32)
    if qr.lib().qr_device_count() == 0:
        with pytest.raises(qr.QrError):
            qr.frame_register(arr)
    else:
        qr.frame_register(arr); qr.frame_unregister(arr)
    with pytest.raises(qr.QrError):
        qr.frame_unregister(arr)                                    # not registered (any more)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_render_host_over_a_device_list(qr, monkeypatch, name, devices):
    blob = load_blob(name)
    scn = qr.Scene(blob)
    want = load_frame(name) & 0xFFFFFF
    monkeypatch.delenv("QR_DEVICES", raising=False)
    assert (scn.render_host() == want).all()
    monkeypatch.setenv("QR_DEVICES", devices)
    assert (scn.render_host() == want).all()
    assert (scn.render_host() == want).all()                        # the replicas of the first call are reused
    scn.set_depth(0)
    monkeypatch.delenv("QR_DEVICES")
    one = scn.render_host()
    monkeypatch.setenv("QR_DEVICES", devices)
    assert (scn.render_host() == one).all()                         # launch parameters travel with every call


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_gpu_registered_frame_is_written_directly(qr, monkeypatch, devices):
    """qr_frame_register: compact and strided frames, whole frames and the row ownership of the engine's threads."""
    name = "demo02_160_gf_aa4"
    scn = qr.Scene(load_blob(name))
    want = load_frame(name) & 0xFFFFFF
    h, w = want.shape
    if devices:
        monkeypatch.setenv("QR_DEVICES", devices)
    else:
        monkeypatch.delenv("QR_DEVICES", raising=False)
    pad = 24
    buf = np.full((h, w + pad), 0xDEADBEEF, dtype=np.uint32)
    qr.frame_register(buf)
    try:
        scn.render_host(out=buf, row_pixels=w + pad)
        assert (buf[:, :w] == want).all() and (buf[:, w:] == 0xDEADBEEF).all()
        # rows 1, 4, 7, ... of rows 16..88 only (qr_scene_set_rows: what a worker thread of the engine owns)
        buf[:] = 0xDEADBEEF
        scn.set_rows(16, 88, 1, 3)
        scn.render_host(out=buf, row_pixels=w + pad)
        own = np.zeros(h, dtype=bool); own[16:88] = (np.arange(16, 88) % 3) == 1
        assert (buf[own, :w] == want[own]).all() and (buf[~own] == 0xDEADBEEF).all() and (buf[:, w:] == 0xDEADBEEF).all()
        scn.set_rows(0, h, 0, 1)
    finally:
        qr.frame_unregister(buf)
    flat = np.zeros((h, w), dtype=np.uint32)
    qr.frame_register(flat)
    try:
        assert (scn.render_host(out=flat) == want).all()
    finally:
        qr.frame_unregister(flat)
    assert (scn.render_host() == want).all()                        # and the staging path afterwards


@pytest.mark.gpu
@pytest.mark.parametrize("env,extra", [({"QR_DEVICES": "0,0"}, []), ({"QR_DEVICES": "0,0,0,0,0,0"}, []), ({}, ["--pin-frame"]),
                                       ({"QR_DEVICES": "0,0,0"}, ["--pin-frame"])])
def test_gpu_drop_in_over_a_device_list_and_a_registered_frame(env, extra):
    """The unmodified engine through qr_render0 (oracle/_ref/qr_ref_shim): several device entries, the engine's frame
    registered by the driver the way the binding would, both; frames equal the engine's own."""
    if not os.path.exists(SHIM):
        pytest.skip("oracle/_ref/qr_ref_shim was not built (needs /root/reference at build time)")
    tmp = tempfile.mkdtemp(prefix="qrdev_")
    os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)
    for args in (["--scene", "demo01", "-w", "640", "-h", "480"],
                 ["--scene", "demo02", "-w", "320", "-h", "240", "--gamma", "--fresnel", "--fsaa", "4", "-t", "3000", "--bench", "3", "--animate", "40"],
                 ["--scene", "demo03", "-w", "333", "-h", "211", "--threads", "4", "--fsaa", "2"],
                 ["--scene", "test13", "-w", "64", "-h", "24"]):
        out = subprocess.run([SHIM] + args + ["--gpu"] + extra, cwd=tmp, capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, QR_VERBOSE="1", **env))
        assert out.returncode == 0, out.stdout + out.stderr
        assert "MATCH" in out.stdout and "MISMATCH" not in out.stdout, out.stdout
        n = len(env.get("QR_DEVICES", "0").split(","))
        assert f"{n} device slot" in out.stderr, out.stderr[-2000:]
        if extra:
            assert "registered frame: direct" in out.stderr, out.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_replicas_do_not_outlive_their_scene(qr, monkeypatch):
    """An animation uploads a scene of the same size every frame: the destroyed scene's address and its image's device
    address are reused by the next upload, so the per-thread QR_DEVICES replicas are keyed on the upload's serial, not on
    pointers and sizes.  Upload A, render over a device list, destroy A, upload B of the same size (A with the camera moved:
    same records and lists, another picture; expected frame from the oracle), render: B's frame, not A's."""
    import struct
    import qr_oracle
    a = load_blob("demo01_160")
    off_frame = struct.unpack_from("<I", a, 40)[0]                   # qr_header.off_frame (include/qr_scene.h)
    org = list(struct.unpack_from("<3f", a, off_frame + 25 * 4))     # qr_frame.org
    b = bytearray(a)
    struct.pack_into("<3f", b, off_frame + 25 * 4, org[0] + 0.75, org[1] - 0.5, org[2] + 0.25)
    b = bytes(b)
    want_a = load_frame("demo01_160") & 0xFFFFFF
    want_b = qr_oracle.render(b, threads=4)[0] & 0xFFFFFF
    assert (want_a != want_b).sum() > 1000
    monkeypatch.setenv("QR_DEVICES", "0,0")
    same_address = 0
    for _ in range(4):
        sa = qr.Scene(a)
        ha = sa._h.value
        assert (sa.render_host() == want_a).all()
        sa.close()
        sb = qr.Scene(b)
        same_address += sb._h.value == ha
        got = sb.render_host()
        assert (got == want_b).all(), f"stale replicas: {(got != want_b).sum()} pixels differ (handle address reused: {sb._h.value == ha})"
        sb.close()
    print(f"scene handle address reused in {same_address} of 4 rounds")


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,7", "0,-1", "0,x", "0,0,0,0,0,0,0,0,0"])
def test_gpu_bad_device_lists_are_refused(qr, monkeypatch, devices):
    """QR_DEVICES: an ordinal the box does not have, a negative one, garbage, more than QR_MAX_DEVICES entries: QR_ERR_ARG,
    not a silently shortened or truncated list (a one-GPU box has ordinal 0 only)."""
    scn = qr.Scene(load_blob("demo01_160"))
    monkeypatch.setenv("QR_DEVICES", devices)
    if qr.lib().qr_device_count() > 7 and devices == "0,7":
        pytest.skip("ordinal 7 exists on this box")
    with pytest.raises(qr.QrError):
        scn.render_host()
    monkeypatch.delenv("QR_DEVICES")
    assert (scn.render_host() == (load_frame("demo01_160") & 0xFFFFFF)).all()
