"""CPU suite: the C-ABI library loads and exports every symbol include/qrhip.h declares
(no compute calls here - there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_blob


def test_library_exports_every_declared_symbol(qr):
    header = open(os.path.join(ROOT, "include", "qrhip.h")).read()
    declared = set(re.findall(r"\b(qr_[a-z0-9_]+)\s*\(", header))
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
