"""What rebuilt tile lists render, pinned against the reference itself.

tests/golden/rebin/ holds every jittered scene of the drop-in fuzz on which the UNMODIFIED reference's frame with its
screen tiling on (engine.cpp:1956-2128, 3129-3253) differs from its own frame with only that optimisation off
(`qr_ref --opts-off tiling`; tests/golden/make_rebin_golden.py): 11 of 36.  Claims pinned here, for all 11:
  1. with the engine's tile lists the oracle (CPU) and the HIP backend (GPU) reproduce the TILED frame;
  2. with the camera list in every tile (oracle) / with tile lists rebuilt by the GPU binning pass (HIP backend) they
     reproduce the TILING-OFF frame, bit for bit -- the binning pass is exact, it is the engine's tiling that is not
     conservative on these transforms;
  3. the two frames differ only where the primary hits differ, and in every such pixel the surface the camera-list walk
     hits is MISSING from the engine's list of that pixel's tile (the tiled frame shows what lies behind it, or nothing):
     the engine's tiling dropped a surface from a tile it covers; it never adds one.
Frames are compared through the manifest's FNV-1a-64 hashes (all cases) and pixel for pixel (the three cases whose
frames are committed).
"""
import gzip
import io
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

REBIN = os.path.join(GOLDEN, "rebin")
with open(os.path.join(REBIN, "manifest.json")) as _f:
    _MAN = json.load(_f)
CASES = _MAN["cases"]
NAMES = sorted(CASES)


def _blob(name):
    with open(os.path.join(REBIN, CASES[name]["snapshot"]), "rb") as f:
        return gzip.decompress(f.read())


def _frame(name, tag):
    with open(os.path.join(REBIN, CASES[name]["frame_" + tag]), "rb") as f:
        return np.load(io.BytesIO(gzip.decompress(f.read())))


def _untiled(blob):
    """The snapshot with ONE tile that spans the frame and holds the camera list: what a walk without tile lists sees."""
    b = bytearray(blob)
    off_frame = struct.unpack_from("<I", b, 4 * 10)[0]
    fi = np.frombuffer(b, dtype=np.int32, count=49, offset=off_frame).copy()
    fi[34], fi[35], fi[36], fi[37] = fi[31], fi[32], 1, 1          # tile_w, tile_h = frm_w, frm_h; one tile
    b[off_frame:off_frame + 196] = fi.tobytes()
    struct.pack_into("<I", b, 4 * 8, 1)                             # n_tiles
    off_tiles = struct.unpack_from("<I", b, 4 * 15)[0]
    struct.pack_into("<i", b, off_tiles, int(fi[38]))               # tiles[0] = clist
    return bytes(b)


def _tile_members(blob):
    """per tile of the snapshot: the set of surface indices in its list (qr_elem chains, include/qr_scene.h)"""
    hdr = struct.unpack_from("<26I", blob, 0)
    n_elm, n_tiles, off_frame, off_elm, off_tiles = hdr[7], hdr[8], hdr[10], hdr[14], hdr[15]
    fi = np.frombuffer(blob, dtype=np.int32, count=49, offset=off_frame)
    elm = np.frombuffer(blob, dtype=np.int32, count=n_elm * 4, offset=off_elm).reshape(n_elm, 4)
    tiles = np.frombuffer(blob, dtype=np.int32, count=n_tiles, offset=off_tiles)
    out = []
    for head in tiles:
        s, e = set(), int(head)
        while e >= 0:
            s.add(int(elm[e, 0])); e = int(elm[e, 2])
        out.append(s)
    return out, int(fi[34]), int(fi[35]), int(fi[36])         # tile_w, tile_h, tls_row


def _check_dropped(blob, ids_tiled, ids_untiled, diff):
    """claim 3 of the module docstring"""
    other = ids_tiled != ids_untiled
    assert not (diff & ~other).any(), "a pixel can only differ where the primary hits differ (tile lists serve primary rays only)"
    members, tw, th, row = _tile_members(blob)
    ys, xs = np.nonzero(other)
    for y, x in zip(ys.tolist(), xs.tolist()):
        hit = int(ids_untiled[y, x])
        assert hit >= 0, "the walk without tiles must hit something the tiled walk does not"
        assert (hit >> 1) not in members[(y // th) * row + x // tw], f"pixel ({x}, {y}): surface {hit >> 1} is in its tile's list"


def test_manifest_is_the_fuzz_population():
    assert len(CASES) == 11 and len(_MAN["scenes_where_tiling_changes_nothing"]) == 25
    assert sum(1 for c in CASES.values() if "frame_tiled" in c) == 3
    for c in CASES.values():
        assert c["hash_tiled"] != c["hash_untiled"] and c["differing_pixels"] > 0


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_both_reference_frames(oracle, name):
    c = CASES[name]
    blob = _blob(name)
    tiled, ids, _ = oracle.render(blob, threads=8, want_ids=True)
    assert oracle.frame_hash(tiled) == int(c["hash_tiled"], 16), "engine's tile lists -> the reference's tiled frame"
    untiled, ids_u, _ = oracle.render(_untiled(blob), threads=8, want_ids=True)
    assert oracle.frame_hash(untiled) == int(c["hash_untiled"], 16), "camera list everywhere -> the reference's tiling-off frame"
    diff = tiled != untiled
    assert int(diff.sum()) == c["differing_pixels"]
    _check_dropped(blob, ids, ids_u, diff)
    if "frame_tiled" in c:
        assert (tiled == _frame(name, "tiled")).all() and (untiled == _frame(name, "untiled")).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [None, "8x8"])
@pytest.mark.parametrize("name", NAMES)
def test_gpu_rebuilt_tile_lists_give_the_tiling_off_frame(qr, oracle, name, tile):
    import torch
    c = CASES[name]
    blob = _blob(name)
    base = qr.Scene(blob)
    f0 = base.new_frame(); i0 = torch.full_like(f0, -2)
    base.render(f0, ids=i0); torch.cuda.synchronize()
    tiled = f0.cpu().numpy().view(np.uint32)
    assert oracle.frame_hash(tiled) == int(c["hash_tiled"], 16)
    if tile:
        os.environ["QR_BIN_TILE"] = tile
    try:
        scn = qr.Scene(blob, rebin_tiles=True)
    finally:
        os.environ.pop("QR_BIN_TILE", None)
    f1 = scn.new_frame(); i1 = torch.full_like(f1, -2)
    scn.render(f1, ids=i1); torch.cuda.synchronize()
    rebuilt = f1.cpu().numpy().view(np.uint32)
    assert oracle.frame_hash(rebuilt) == int(c["hash_untiled"], 16), "rebuilt tile lists must give the reference's tiling-off frame"
    diff = tiled != rebuilt
    assert int(diff.sum()) == c["differing_pixels"]
    _check_dropped(blob, i0.cpu().numpy(), i1.cpu().numpy(), diff)
    if "frame_untiled" in c:
        assert (rebuilt == _frame(name, "untiled")).all() and (tiled == _frame(name, "tiled")).all()
