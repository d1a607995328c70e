"""quadray-engine_amd: Python binding of libqrhip.so (the C ABI in include/qrhip.h).

The directory name is not a valid Python identifier, so import it by path
(tests/conftest.py, bench.py and __graft_entry__.py use `load_package()` from
the repository root helper `qr_loader.py`).

PyTorch is used only for device memory, streams and torch.distributed; every
pixel is computed by the hand-written HIP kernel behind the C ABI.  There is
no CPU fallback: without the shared library or without a GPU these calls raise.
"""
import ctypes
import gzip
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QR_LIB") or os.path.join(_HERE, "libqrhip.so")   # QR_LIB: A/B experiment builds

# every symbol include/qrhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "qr_render0", "qr_capture_snapshot", "qr_flatten", "qr_free",
    "qr_scene_upload", "qr_scene_upload_ex", "qr_program_stats", "qr_snapshot_build_lists_c", "qr_scene_destroy", "qr_scene_get_info", "qr_scene_set_depth", "qr_scene_set_pt",
    "qr_scene_set_rows", "qr_scene_set_tile_rows", "qr_render_async", "qr_render_multi_async", "qr_render_ids_async",
    "qr_render_count", "qr_render_host", "qr_render_timed",
    "qr_frame_register", "qr_frame_unregister",
    "qr_frame_hash", "qr_last_error", "qr_version", "qr_device_count", "qr_kernel_name", "qr_capture_index",
    # include/qr_hierarchy.h
    "qr_hierarchy_update", "qr_hierarchy_animate", "qr_hierarchy_apply", "qr_hierarchy_bounds", "qr_anim_spin", "qr_anim_swing",
]


UPLOAD_REBIN_TILES = 1


class QrError(RuntimeError):
    pass


class SceneInfo(ctypes.Structure):
    _fields_ = [("frm_w", ctypes.c_int32), ("frm_h", ctypes.c_int32), ("fsaa", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("n_srf", ctypes.c_int32), ("n_mat", ctypes.c_int32),
                ("n_lgt", ctypes.c_int32), ("n_elm", ctypes.c_int32), ("n_tiles", ctypes.c_int32),
                ("n_texels", ctypes.c_int32), ("tile_w", ctypes.c_int32), ("tile_h", ctypes.c_int32),
                ("device_bytes", ctypes.c_uint64)]


class ProgramInfo(ctypes.Structure):
    _fields_ = [("bytes", ctypes.c_uint64), ("n_lists", ctypes.c_uint32), ("n_cells", ctypes.c_uint32),
                ("n_dropped", ctypes.c_uint32), ("n_clip_cells", ctypes.c_uint32), ("n_sched", ctypes.c_uint32),
                ("n_grids", ctypes.c_uint32), ("n_grid_lists", ctypes.c_uint32), ("n_dda", ctypes.c_uint32)]


class RayCounts(ctypes.Structure):
    _fields_ = [("primary", ctypes.c_uint64), ("shadow", ctypes.c_uint64),
                ("reflect", ctypes.c_uint64), ("refract", ctypes.c_uint64)]

    def total(self):
        return self.primary + self.shadow + self.reflect + self.refract

    def as_dict(self):
        return dict(primary=self.primary, shadow=self.shadow, reflect=self.reflect, refract=self.refract)


_lib = None


def lib():
    """Load libqrhip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(the gfx950 backend has no fallback path)")
    try:
        # load torch's bundled HIP runtime first so that libqrhip.so binds to the same libamdhip64
        # (two HIP runtimes in one process cannot both own the device)
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cu64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
    L.qr_last_error.restype = ctypes.c_char_p
    L.qr_version.restype = ctypes.c_char_p
    L.qr_kernel_name.restype = ctypes.c_char_p
    L.qr_device_count.restype = ci
    L.qr_scene_upload.argtypes = [vp, cu64, ci, ctypes.POINTER(vp)]
    L.qr_scene_upload_ex.argtypes = [vp, cu64, ci, ctypes.c_uint32, ctypes.POINTER(vp)]
    L.qr_scene_destroy.argtypes = [vp]
    L.qr_program_stats.argtypes = [vp, cu64, ctypes.POINTER(ProgramInfo)]
    L.qr_snapshot_build_lists_c.argtypes = [vp, cu64, ctypes.POINTER(vp), ctypes.POINTER(cu64)]
    L.qr_free.argtypes = [vp]
    L.qr_frame_hash.argtypes = [vp, cu64]
    L.qr_frame_hash.restype = cu64
    L.qr_scene_get_info.argtypes = [vp, ctypes.POINTER(SceneInfo)]
    L.qr_scene_set_depth.argtypes = [vp, ci]
    L.qr_scene_set_pt.argtypes = [vp, ci]
    L.qr_scene_set_rows.argtypes = [vp, ci, ci, ci, ci]
    L.qr_scene_set_tile_rows.argtypes = [vp, ci, ci]
    L.qr_render_async.argtypes = [vp, vp, vp]
    L.qr_render_ids_async.argtypes = [vp, vp, vp, vp]
    L.qr_render_multi_async.argtypes = [ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ci), ctypes.POINTER(ci), vp]
    L.qr_render_count.argtypes = [vp, vp, vp, ctypes.POINTER(RayCounts)]
    L.qr_render_host.argtypes = [vp, vp, ci]
    L.qr_frame_register.argtypes = [vp, ctypes.c_uint64]
    L.qr_frame_unregister.argtypes = [vp]
    L.qr_render_timed.argtypes = [vp, vp, vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.qr_hierarchy_update.argtypes = [vp, ci, ctypes.c_uint32, vp]
    L.qr_hierarchy_bounds.argtypes = [vp, cu64, vp, ci, ctypes.c_uint32, vp]
    L.qr_hierarchy_animate.argtypes = [vp, ci, ctypes.c_int64, vp, vp, vp, ci]
    L.qr_hierarchy_apply.argtypes = [vp, cu64, vp, vp, ci, ctypes.c_uint32, ci, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(cu64)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise QrError(f"qrhip error {rc}: {lib().qr_last_error().decode()}")


def read_snapshot(path):
    """Return the raw snapshot bytes of a .qrs or .qrs.gz file."""
    with open(path, "rb") as f:
        raw = f.read()
    return gzip.decompress(raw) if path.endswith(".gz") else raw


def frame_register(arr):
    """qr_frame_register on a numpy array the caller keeps alive until frame_unregister(arr)."""
    _check(lib().qr_frame_register(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))


def frame_unregister(arr):
    _check(lib().qr_frame_unregister(arr.ctypes.data_as(ctypes.c_void_p)))


def frame_hash(frame):
    """FNV-1a-64 fingerprint of a frame (numpy uint32 array or CUDA int32 tensor), as in tests/golden/manifest.json."""
    import numpy as np
    if hasattr(frame, "cpu"):
        frame = frame.cpu().numpy()
    f = np.ascontiguousarray(frame).view(np.uint32)
    return int(lib().qr_frame_hash(f.ctypes.data_as(ctypes.c_void_p), f.size))


def build_lists(blob):
    """Per-surface shadow / reflection / light lists from the global list (host pass, no GPU): returns a new snapshot."""
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    buf = ctypes.create_string_buffer(blob, len(blob))
    _check(lib().qr_snapshot_build_lists_c(buf, len(blob), ctypes.byref(out), ctypes.byref(n)))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        lib().qr_free(out)


# ---- object hierarchy (include/qr_hierarchy.h): numpy record arrays in the C layout of qr_node / qr_node_state ----

def node_dtype():
    import numpy as np
    return np.dtype([("parent", "<i4"), ("tag", "<i4"), ("scl", "<f4", 3), ("rot", "<f4", 3), ("pos", "<f4", 3),
                     ("shape", "<f4", 3), ("srf", "<i4"), ("inb", "<i4"), ("bvb", "<i4"), ("lgt", "<i4"), ("anim", "<i4"),
                     ("pov", "<f4"), ("bvnode", "<i4"), ("nverts", "<i4"), ("lmin", "<f4", 3), ("lmax", "<f4", 3),
                     ("tex", "<f4", 8), ("has_tex", "<i4"), ("pad_", "<i4", 3)])


def node_bounds_dtype():
    import numpy as np
    return np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("cmin", "<f4", 3), ("cmax", "<f4", 3), ("mid", "<f4", 3), ("rad", "<f4"),
                     ("nverts", "<i4"), ("inmin", "<f4", 3), ("inmax", "<f4", 3), ("inmid", "<f4", 3), ("inrad", "<f4"),
                     ("trmin", "<f4", 3), ("trmax", "<f4", 3), ("trrad", "<f4"), ("inb_form", "<i4"), ("bvb_form", "<i4")])


def node_state_dtype():
    import numpy as np
    return np.dtype([("mtx", "<f4", 16), ("map", "<i4", 4), ("sgn", "<i4", 4), ("scl", "<f4", 4), ("trnode", "<i4"),
                     ("obj_has_trm", "<i4"), ("mtx_has_trm", "<i4"), ("pad", "<i4")])


HIER_RESET_TILES = 1
HIER_BOUNDS = 2
HIER_REGROUP = 4
ANIM_SPIN, ANIM_SWING = "spin", "swing"


def hierarchy_update(nodes, opts):
    """Hierarchical transform update (qr_hierarchy_update): nodes (node_dtype array) -> node_state_dtype array."""
    import numpy as np
    nodes = np.ascontiguousarray(nodes, dtype=node_dtype())
    out = np.zeros(len(nodes), dtype=node_state_dtype())
    _check(lib().qr_hierarchy_update(nodes.ctypes.data_as(ctypes.c_void_p), len(nodes), opts, out.ctypes.data_as(ctypes.c_void_p)))
    return out


def hierarchy_bounds(blob, nodes, opts):
    """Bounding and clipping boxes of every node (qr_hierarchy_bounds): node_bounds_dtype array."""
    import numpy as np
    nodes = np.ascontiguousarray(nodes, dtype=node_dtype())
    out = np.zeros(len(nodes), dtype=node_bounds_dtype())
    buf = ctypes.create_string_buffer(blob, len(blob))
    _check(lib().qr_hierarchy_bounds(buf, len(blob), nodes.ctypes.data_as(ctypes.c_void_p), len(nodes), opts, out.ctypes.data_as(ctypes.c_void_p)))
    return out


class _AnimParams(ctypes.Structure):
    _fields_ = [("axis", ctypes.c_int32), ("rate", ctypes.c_float), ("period", ctypes.c_float), ("pad", ctypes.c_int32)]


def hierarchy_animate(nodes, time, node_time, animators):
    """Run the animators (qr_hierarchy_animate) in place on nodes / node_time (int64 array, -1 = never updated).
    animators[k] is (ANIM_SPIN, axis, rate), (ANIM_SWING, axis, rate, period) or a Python callable
    f(time, last_time, trm) with trm a 9-float numpy view (scl, rot, pos) it changes in place."""
    import numpy as np
    assert nodes.dtype == node_dtype() and nodes.flags["C_CONTIGUOUS"] and node_time.dtype == np.int64
    proto = ctypes.CFUNCTYPE(None, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p)
    n = len(animators)
    fns, users, keep = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_void_p * max(n, 1))(), []
    for k, a in enumerate(animators):
        if callable(a):
            cb = proto(lambda t, lt, trm, _u, a=a: a(t, lt, np.ctypeslib.as_array(trm, shape=(9,))))
            keep.append(cb)
            fns[k] = ctypes.cast(cb, ctypes.c_void_p).value
        else:
            prm = _AnimParams(int(a[1]), float(a[2]), float(a[3]) if len(a) > 3 else 0.0, 0)
            keep.append(prm)
            fns[k] = ctypes.cast(lib().qr_anim_spin if a[0] == ANIM_SPIN else lib().qr_anim_swing, ctypes.c_void_p).value
            users[k] = ctypes.addressof(prm)
    _check(lib().qr_hierarchy_animate(nodes.ctypes.data_as(ctypes.c_void_p), len(nodes), int(time),
                                      node_time.ctypes.data_as(ctypes.c_void_p), fns, users, n))


def hierarchy_apply(blob, nodes, opts, camera=-1, base=None, flags=0):
    """Write the transform fields the nodes imply into a copy of the snapshot (qr_hierarchy_apply); base: the nodes the
    snapshot was captured with (enables the scope checks).  Rebuild the lists afterwards (build_lists)."""
    import numpy as np
    nodes = np.ascontiguousarray(nodes, dtype=node_dtype())
    bp = None
    if base is not None:
        base = np.ascontiguousarray(base, dtype=node_dtype())
        assert len(base) == len(nodes)
        bp = base.ctypes.data_as(ctypes.c_void_p)
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    buf = ctypes.create_string_buffer(blob, len(blob))
    _check(lib().qr_hierarchy_apply(buf, len(blob), bp, nodes.ctypes.data_as(ctypes.c_void_p), len(nodes), opts, camera, flags,
                                    ctypes.byref(out), ctypes.byref(n)))
    try:
        return ctypes.string_at(out, n.value)
    finally:
        lib().qr_free(out)


def hierarchy_records_after_apply(blob, nodes, opts):
    """The node table as it stands after hierarchy_apply(blob, nodes, ...): an array that became the transform node of surfaces
    and had no record got one -- the k-th such array in node order holds record n_srf + k (include/qr_hierarchy.h).  Use it
    as `base` (and for `srf` of the next table) when the patched snapshot is patched again."""
    import struct
    import numpy as np
    nodes = np.ascontiguousarray(nodes, dtype=node_dtype()).copy()
    st = hierarchy_update(nodes, opts)
    n_srf = struct.unpack_from("<I", blob, 16)[0]
    heads = set()
    for i in range(len(nodes)):
        t = int(st[i]["trnode"])
        if 0 <= nodes[i]["tag"] < 9 and nodes[i]["srf"] >= 0 and t >= 0 and t != i:
            heads.add(t)
    for g in sorted(heads):
        if nodes[g]["srf"] < 0:
            nodes[g]["srf"] = n_srf
            n_srf += 1
    return nodes


def program_stats(blob):
    """Validate + compile a snapshot on the host (no GPU): the device image's size and cell counts."""
    info = ProgramInfo()
    buf = ctypes.create_string_buffer(blob, len(blob))
    _check(lib().qr_program_stats(buf, len(blob), ctypes.byref(info)))
    return info


class Scene:
    """A snapshot resident on one GPU (qr_device_scene)."""

    def __init__(self, blob, device=0, rebin_tiles=False):
        """rebin_tiles: rebuild the per-tile lists on the GPU from the camera list
        (QR_UPLOAD_REBIN_TILES, include/qrhip.h) instead of using the snapshot's."""
        self._h = ctypes.c_void_p()
        self._buf = ctypes.create_string_buffer(blob, len(blob))
        _check(lib().qr_scene_upload_ex(self._buf, len(blob), device, UPLOAD_REBIN_TILES if rebin_tiles else 0,
                                        ctypes.byref(self._h)))
        self.device = device
        self.info = SceneInfo()
        _check(lib().qr_scene_get_info(self._h, ctypes.byref(self.info)))

    @property
    def width(self):
        return self.info.frm_w

    @property
    def height(self):
        return self.info.frm_h

    def close(self):
        if self._h:
            lib().qr_scene_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_depth(self, depth):
        _check(lib().qr_scene_set_depth(self._h, depth))
        self.info.depth = depth

    def set_pt(self, on=True, eager=False):
        """Path-tracer mode: every render() then adds one sample per pixel sample; the frame is the running mean.
        eager: shade in the reference's order (every hit that passes the depth test, at once): its random streams."""
        _check(lib().qr_scene_set_pt(self._h, (2 if eager else 1) if on else 0))

    def set_rows(self, row_begin, row_end, index=0, thnum=1):
        _check(lib().qr_scene_set_rows(self._h, row_begin, row_end, index, thnum))

    def set_tile_rows(self, first, stride):
        _check(lib().qr_scene_set_tile_rows(self._h, first, stride))

    def new_frame(self):
        import torch
        return torch.zeros((self.height, self.width), dtype=torch.int32, device=f"cuda:{self.device}")

    @staticmethod
    def _stream_ptr(stream):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        return ctypes.c_void_p(s.cuda_stream)

    def render(self, frame=None, stream=None, ids=None):
        """Asynchronous launch on `stream` (default: torch's current stream)."""
        if frame is None:
            frame = self.new_frame()
        sp = self._stream_ptr(stream)
        if ids is None:
            _check(lib().qr_render_async(self._h, ctypes.c_void_p(frame.data_ptr()), sp))
        else:
            _check(lib().qr_render_ids_async(self._h, ctypes.c_void_p(frame.data_ptr()),
                                             ctypes.c_void_p(ids.data_ptr()), sp))
        return frame

    def render_count(self, frame=None, stream=None):
        if frame is None:
            frame = self.new_frame()
        c = RayCounts()
        _check(lib().qr_render_count(self._h, ctypes.c_void_p(frame.data_ptr()), self._stream_ptr(stream), ctypes.byref(c)))
        return frame, c

    def render_timed(self, frame, iters, stream=None):
        avg, mn = ctypes.c_float(), ctypes.c_float()
        _check(lib().qr_render_timed(self._h, ctypes.c_void_p(frame.data_ptr()), self._stream_ptr(stream),
                                     iters, ctypes.byref(avg), ctypes.byref(mn)))
        return avg.value, mn.value

    def render_host(self, out=None, row_pixels=None):
        """qr_render_host into a new (or the given) host frame; `row_pixels`: its stride in pixels."""
        import numpy as np
        if out is None:
            out = np.zeros((self.height, self.width), dtype=np.uint32)
        _check(lib().qr_render_host(self._h, out.ctypes.data_as(ctypes.c_void_p), self.width if row_pixels is None else row_pixels))
        return out


class MultiRender:
    """Prepared multi-target launch (qr_render_multi_async): targets = [(scene, frame tensor, row_begin, row_end)].
    The ctypes argument arrays are built once; call it with a stream to launch."""

    def __init__(self, targets):
        n = len(targets)
        self.n = n
        self._keep = targets
        self._scenes = (ctypes.c_void_p * n)(*[t[0]._h.value for t in targets])
        self._frames = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in targets])
        self._r0 = (ctypes.c_int * n)(*[int(t[2]) for t in targets])
        self._r1 = (ctypes.c_int * n)(*[int(t[3]) for t in targets])

    def __call__(self, stream=None):
        _check(lib().qr_render_multi_async(self.n, self._scenes, self._frames, self._r0, self._r1, Scene._stream_ptr(stream)))
