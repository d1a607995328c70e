"""TEST INFRASTRUCTURE: ctypes binding of oracle/libqr_oracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqr_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        L.qro_render.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p]
        L.qro_render.restype = ctypes.c_int
        L.qro_render2.argtypes = L.qro_render.argtypes + [ctypes.c_int]
        L.qro_render2.restype = ctypes.c_int
        L.qro_last_flops.restype = ctypes.c_uint64
        L.qro_info.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.qro_info.restype = ctypes.c_int
        L.qro_hash.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.qro_hash.restype = ctypes.c_uint64
        _lib = L
    return _lib


def info(blob):
    out = (ctypes.c_int32 * 8)()
    buf = ctypes.create_string_buffer(blob, len(blob))
    rc = lib().qro_info(buf, len(blob), out)
    if rc != 0:
        raise RuntimeError(f"qro_info rc={rc}")
    return dict(w=out[0], h=out[1], fsaa=out[2], depth=out[3], n_srf=out[4], n_elm=out[5], index=out[6], thnum=out[7])


def render(blob, depth=-1, threads=0, want_ids=False, rows=None, index=0, thnum=1, deferred=False):
    """Render a snapshot on the CPU. Returns (frame uint32 [h,w], ids or None, counts dict).
    deferred=False is the reference's semantics (every depth-test winner is shaded);
    deferred=True shades only the final hit (same pixels, the HIP backend's ray count)."""
    i = info(blob)
    w, h = i["w"], i["h"]
    frame = np.zeros((h, w), dtype=np.uint32)
    ids = np.full((h, w), -1, dtype=np.int32) if want_ids else None
    counts = (ctypes.c_uint64 * 4)()
    buf = ctypes.create_string_buffer(blob, len(blob))
    r0, r1 = rows if rows is not None else (0, h)
    rc = lib().qro_render2(buf, len(blob), frame.ctypes.data, ids.ctypes.data if want_ids else None,
                           depth, r0, r1, index, thnum, threads, counts, 1 if deferred else 0)
    if rc != 0:
        raise RuntimeError(f"qro_render rc={rc}")
    return frame, ids, dict(primary=counts[0], shadow=counts[1], reflect=counts[2], refract=counts[3],
                            flops=int(lib().qro_last_flops()))


def frame_hash(frame):
    f = np.ascontiguousarray(frame, dtype=np.uint32)
    return int(lib().qro_hash(f.ctypes.data, f.size))
