#!/usr/bin/env python3
"""Render a scene with the unmodified reference (oracle/_ref) and with our CPU restatement
(oracle/libqr_oracle.so) on the captured snapshot; report pixel differences.
Usage: oracle_check.py SCENE W H [--fsaa N] [--gamma] [--fresnel] [--depth D] [-t MS] [--opts none|full]"""
import ctypes, subprocess, sys, os, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")

def load_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libqr_oracle.so"))
    lib.qro_render.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_void_p]
    lib.qro_render.restype = ctypes.c_int
    lib.qro_hash.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.qro_hash.restype = ctypes.c_uint64
    return lib

def oracle_render(lib, blob, w, h, depth=-1, threads=8, want_ids=False):
    frame = np.zeros((h, w), dtype=np.uint32)
    ids = np.zeros((h, w), dtype=np.int32) if want_ids else None
    counts = (ctypes.c_uint64 * 4)()
    buf = ctypes.create_string_buffer(blob, len(blob))
    rc = lib.qro_render(buf, len(blob), frame.ctypes.data, ids.ctypes.data if want_ids else None,
                        depth, 0, h, 0, 1, threads, counts)
    if rc != 0:
        raise RuntimeError(f"qro_render rc={rc}")
    return frame, ids, list(counts)

def ref_render(scene, w, h, extra, want_snapshot=True):
    tmp = tempfile.mkdtemp(prefix="qrchk_")
    raw = os.path.join(tmp, "f.raw"); qrs = os.path.join(tmp, "s.qrs")
    os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)
    cmd = [os.path.join(REF, "qr_ref_shim"), "--scene", scene, "-w", str(w), "-h", str(h), "--out", raw] + extra
    if want_snapshot:
        cmd += ["--snapshot", qrs]
    out = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stdout + out.stderr)
    frame = np.fromfile(raw, dtype=np.uint32).reshape(h, w)
    blob = open(qrs, "rb").read() if want_snapshot else None
    return frame, blob, out.stdout

def main():
    scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    extra = sys.argv[4:]
    lib = load_oracle()
    ref, blob, log = ref_render(scene, w, h, extra)
    ours, ids, counts = oracle_render(lib, blob, w, h)
    ref = ref & 0xFFFFFF
    diff = ref != ours
    n = int(diff.sum())
    print(f"{scene} {w}x{h} {' '.join(extra)}: differing pixels {n} / {w*h}  rays {counts}")
    if n:
        ys, xs = np.nonzero(diff)
        d = np.abs(((ref[diff][:, None] >> np.array([16, 8, 0])) & 255).astype(int) -
                   ((ours[diff][:, None] >> np.array([16, 8, 0])) & 255).astype(int))
        print("  max channel delta", d.max(), " first:", [(int(x), int(y), hex(int(ref[y, x])), hex(int(ours[y, x]))) for x, y in list(zip(xs, ys))[:8]])
    return 0 if n == 0 else 1

if __name__ == "__main__":
    sys.exit(main())
