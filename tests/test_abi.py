"""CPU suite: the C-ABI library loads and exports every symbol include/qrhip.h and include/qr_hierarchy.h declare
(no compute calls here - there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_blob


def test_library_exports_every_declared_symbol(qr):
    header = open(os.path.join(ROOT, "include", "qrhip.h")).read() + open(os.path.join(ROOT, "include", "qr_hierarchy.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)            # prose in comments mentions functions of other files
    declared = set(re.findall(r"\b(qr_[a-z0-9_]+)\s*\(", header)) - {"qr_scene_view_init"}
    L = qr.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libqrhip.so lacks {missing}"
    assert set(qr.ABI_SYMBOLS) <= declared


def test_version_and_kernel_name(qr):
    L = qr.lib()
    assert b"gfx950" in L.qr_version()
    assert L.qr_kernel_name() == b"qr_render_kernel"


def test_upload_rejects_malformed_snapshot(qr):
    blob = bytearray(load_blob("demo01_160"))
    blob[0] ^= 0xFF                                   # break the magic
    with pytest.raises(qr.QrError):
        qr.Scene(bytes(blob))


def test_upload_rejects_out_of_range_index(qr):
    import struct
    blob = bytearray(load_blob("demo01_160"))
    off_tiles = struct.unpack_from("<I", blob, 4 * 15)[0]
    struct.pack_into("<i", blob, off_tiles, 1 << 30)  # tile head far outside the element array
    with pytest.raises(qr.QrError):
        qr.Scene(bytes(blob))


def test_no_gpu_fails_loudly(qr):
    """Without a usable HIP device the product path raises; it never falls back to CPU code."""
    if qr.lib().qr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(qr.QrError):
        qr.Scene(load_blob("demo01_160"))


def test_flatten_rejects_bad_abi(qr):
    class Abi(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_uint32), ("quads", ctypes.c_uint32), ("pointer_bits", ctypes.c_uint32),
                    ("address_bits", ctypes.c_uint32), ("element_bits", ctypes.c_uint32), ("endian", ctypes.c_uint32),
                    ("reserved", ctypes.c_uint32 * 2)]
    L = qr.lib()
    abi = Abi(ctypes.sizeof(Abi), 8, 64, 64, 64, 0)   # fp64 build: unsupported
    dummy = ctypes.create_string_buffer(8192)
    blob = ctypes.c_void_p(); size = ctypes.c_uint64()
    rc = L.qr_flatten(dummy, ctypes.byref(abi), ctypes.byref(blob), ctypes.byref(size))
    assert rc == -2


def test_upload_rejects_oversized_frame_and_broken_lists(qr):
    """Limits are checked on the host before anything touches a device: a frame wider than the wave
    schedule can address, a cyclic list and an array that ends outside its list are refused with an
    argument error (also on a machine without a GPU, where a well-formed scene fails with a device error)."""
    import importlib.util, struct
    spec = importlib.util.spec_from_file_location("qr_synth", os.path.join(ROOT, "quadray-engine_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
    L = qr.lib()

    def err_of(blob):
        h = ctypes.c_void_p(); buf = ctypes.create_string_buffer(blob, len(blob))
        rc = L.qr_scene_upload(buf, len(blob), 0, ctypes.byref(h))
        if rc == 0:
            L.qr_scene_destroy(h)
        return rc, L.qr_last_error().decode()

    rc, msg = err_of(synth.make_scene(n_objects=8, width=8 * 16384 + 8, height=8, depth=1, box=8.0))
    assert rc == -1 and "too large" in msg
    good = bytearray(synth.make_scene(n_objects=8, width=64, height=48, depth=1, box=8.0))
    off_elm = struct.unpack_from("<I", good, 4 * 14)[0]
    cyc = bytearray(good)
    off_frame = struct.unpack_from("<I", cyc, 4 * 10)[0]
    head = struct.unpack_from("<i", cyc, off_frame + 4 * 38)[0]             # qr_frame.clist
    e = head
    while struct.unpack_from("<i", cyc, off_elm + 16 * e + 8)[0] != -1:
        e = struct.unpack_from("<i", cyc, off_elm + 16 * e + 8)[0]
    struct.pack_into("<i", cyc, off_elm + 16 * e + 8, head)                 # last cell's next -> head
    rc, msg = err_of(bytes(cyc))
    assert rc == -1 and "cyclic" in msg, msg


def _patch_frame(blob, **fields):
    """A copy of the snapshot with int32 fields of its qr_frame record replaced (include/qr_scene.h)."""
    import struct
    idx = dict(depth=29, fsaa=30, frm_w=31, frm_h=32, tile_w=34, tile_h=35, tls_row=36, tls_col=37, index=39, thnum=40)
    b = bytearray(blob)
    off_frame = struct.unpack_from("<I", b, 4 * 10)[0]
    for k, v in fields.items():
        struct.pack_into("<i", b, off_frame + 4 * idx[k], v)
    return bytes(b)


@pytest.mark.parametrize("fields,what", [
    (dict(depth=-1), "depth"),
    (dict(tls_row=-5, tls_col=-3), "tile grid"),          # product still equals n_tiles = 15
    (dict(tls_row=3, tls_col=5), "cover"),                # 15 tiles of 32x8 cannot cover 160x120
    (dict(index=4, thnum=4), "index"),
    (dict(index=-1), "index"),
    (dict(tile_w=0), "frame parameters"),
])
def test_upload_rejects_bad_frame_records(qr, fields, what):
    """Every frame-record value the kernel indexes with is checked on the host: a negative recursion depth would
    walk the per-lane frame stack out of bounds, a tile grid that does not cover the frame the tile array."""
    blob = load_blob("demo01_160")
    n_tiles = qr.program_stats(blob).n_sched      # well-formed original compiles
    assert n_tiles > 0
    if "tls_row" in fields and fields["tls_row"] > 0:
        import struct
        b = bytearray(_patch_frame(blob, **fields))
        struct.pack_into("<I", b, 4 * 8, fields["tls_row"] * fields["tls_col"])     # keep n_tiles consistent
        bad = bytes(b)
    else:
        bad = _patch_frame(blob, **fields)
    with pytest.raises(qr.QrError) as e:
        qr.program_stats(bad)
    assert what in str(e.value) or "malformed" in str(e.value), str(e.value)
    h = ctypes.c_void_p(); buf = ctypes.create_string_buffer(bad, len(bad))
    assert qr.lib().qr_scene_upload(buf, len(bad), 0, ctypes.byref(h)) == -1       # QR_ERR_ARG before any device is touched


def test_compiler_accepts_every_fixture_and_reports_sizes(qr):
    from conftest import MANIFEST
    for name in sorted(MANIFEST):
        info = qr.program_stats(load_blob(name))
        assert info.n_lists > 0 and info.n_cells > 0 and info.bytes > 4096 and info.n_sched > 0, name
        assert info.bytes % 64 == 0


def test_compiler_survives_corrupted_snapshots(qr):
    """Robustness of validation + list compilation + image verification: random corruption of the element, surface
    and tile arrays either compiles into a verified image or is rejected with an error code -- never a crash,
    never an image with an offset outside itself (qr_program_verify runs on every successful build)."""
    import struct
    import numpy as np
    base = load_blob("demo02_160")
    hdr = struct.unpack_from("<22I", base, 0)
    n_srf, n_elm, n_tiles = hdr[4], hdr[7], hdr[8]
    off_srf, off_elm, off_tiles = hdr[11], hdr[14], hdr[15]
    rng = np.random.default_rng(2024)
    L = qr.lib()
    outcomes = {0: 0, -1: 0, -3: 0}
    for trial in range(400):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 4))
            val = int(rng.choice([-1, 0, 1, 2, 7, n_elm - 1, n_elm, n_srf - 1, n_srf, -2, 1 << 20, int(rng.integers(0, n_elm))]))
            if kind == 0:       # an element field (simd, data, next, kind)
                struct.pack_into("<i", b, off_elm + 16 * int(rng.integers(0, n_elm)) + 4 * int(rng.integers(0, 4)), val)
            elif kind == 1:     # a tag / list-head / trnode field of a surface
                fld = int(rng.choice([7, 11, 15, 19, 23, 34, 35, 36, 37, 38, 39, 40, 41, 44, 45, 46, 47]))
                struct.pack_into("<i", b, off_srf + 256 * int(rng.integers(0, n_srf)) + 4 * fld, val)
            elif kind == 2:     # a tile head
                struct.pack_into("<i", b, off_tiles + 4 * int(rng.integers(0, n_tiles)), val)
            else:               # a material's texture addressing
                off_mat = hdr[12]
                struct.pack_into("<i", b, off_mat + 128 * int(rng.integers(0, hdr[5])) + 4 * int(rng.integers(4, 10)), val)
        info = qr.ProgramInfo()
        buf = ctypes.create_string_buffer(bytes(b), len(b))
        rc = L.qr_program_stats(buf, len(b), ctypes.byref(info))
        assert rc in outcomes, (trial, rc, L.qr_last_error())
        outcomes[rc] += 1
    assert outcomes[0] > 0 and outcomes[-1] > 0, outcomes


@pytest.mark.parametrize("name", ["demo01_160", "demo02_160_gf_aa4", "test13_160"])
def test_walker_reproduces_golden_snapshot(name, tmp_path):
    """The flattener (product code, csrc/qr_walker.cpp) run inside the unmodified reference engine through the
    drop-in shim writes byte for byte the snapshot committed under tests/golden/ (needs oracle/_ref, which
    exists in the build container and travels to the GPU box)."""
    import json, subprocess
    shim = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
    if not os.path.exists(shim):
        pytest.skip("oracle/_ref/qr_ref_shim is not built")
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))[name]
    (tmp_path / "dump").mkdir()
    out = tmp_path / "s.qrs"
    cmd = [shim, "--scene", man["scene"], "-w", str(man["w"]), "-h", str(man["h"]), "--snapshot", str(out)] + man["args"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True, timeout=120)
    assert out.read_bytes() == load_blob(name)
