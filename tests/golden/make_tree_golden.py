#!/usr/bin/env python3
"""Generate the object-hierarchy fixtures under tests/golden/tree/ (run in the build container only).

For every small case of manifest.json the UNMODIFIED reference (oracle/_ref/qr_ref_shim) renders the frame again and
dumps its object tree (oracle/ref_driver.cpp --tree): per object the inputs of the hierarchical update (parent, tag,
scale / rotation / position after the animators ran, shape parameters) and what the engine computed from them (matrix,
transform node, flags), plus the snapshot index of every record the object owns.  The snapshot of that run must equal the
committed one byte for byte, so the indices are valid for it.
    tree/<case>.json.gz        the tree of an existing case
    tree/<extra>.json.gz + tree/<extra>.frame.npy.gz   further animation times of the demo scenes: tree and reference frame
                               only (tests patch the snapshot of another time of the same scene and must get this frame)
Data only: numbers the engine computed; nothing of the reference's source text is stored.
"""
import gzip, io, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
OUT = os.path.join(HERE, "tree")

# name -> (scene, args): frames at other times, same options as an existing snapshot of the scene
EXTRA = {
    "demo01_160_t2500": ("demo01", ["-t", "2500"]),                             # base: demo01_160_t12345
    "demo01_160_gf_t2500": ("demo01", ["--gamma", "--fresnel", "-t", "2500"]),  # base: demo01_160_gf_t5000
    "demo03_160_t3000": ("demo03", ["-t", "3000"]),                             # base: demo03_160 (camera animator only)
    # round 4: targets for the t = 0 snapshots -- the light's array starts rotating there, i.e. the SET of transform nodes changes
    "demo02_160_t4000": ("demo02", ["-t", "4000"]),                             # base: demo02_160
    "demo03_160_t7000": ("demo03", ["-t", "7000"]),                             # base: demo03_160
}


# Every transform of a scene drawn anew (--jitter SEED) with the engine's screen tiling off (its tiling is not conservative on
# such transforms, tests/test_rebin_pin.py): tree + the reference's frame.  Tests patch the t = 0 snapshot of the scene (default
# options) with this tree -- surfaces change the array that is their transform node, become their own, stop; nested arrays,
# bounding volumes and custom clipping included -- and must get this frame.
JITTER = [("demo01", 1), ("demo02", 3), ("demo03", 3), ("test02", 23), ("test03", 22), ("test11", 23), ("test12", 21),
          ("test13", 1), ("test14", 1), ("test16", 3)]


# fuzzed transforms (oracle/ref_driver.cpp --jitter SEED: right-angle and arbitrary rotations, negative and non-unit scalers,
# shifted positions on every object of the scene's static description): tree + the snapshot of that run with its texels
# zeroed (the hierarchy tests do not look at them; 160 KB of texture would travel with every case otherwise)
FUZZ = [("test14", 1, []), ("test14", 2, []), ("test14", 3, ["--opts", "none"]), ("test14", 4, []),
        ("test13", 1, []), ("test16", 1, []), ("demo02", 1, []), ("test12", 1, ["--opts", "none"]), ("test09", 5, []),
        # seed 0: no jitter, the extra arguments make the scene (a swarm of mixed quadrics under rotated groups, then jittered too)
        ("demo03", 0, ["--swarm", "96,5,1"]), ("test09", 7, ["--swarm", "60,6,1"])]


# scenes "a moment later" (oracle/ref_driver.cpp --shift SEED: two thirds of the objects moved by up to 0.5 along every axis,
# nothing else of a transform changed): pairs (A, B) of the same scene.  Kept per state: tree, snapshot (texels zeroed) and, for
# B, the reference's frame.  Tests patch A's snapshot with B's transforms (qr_hierarchy_apply + QR_HIER_BOUNDS): every record
# field must equal B's snapshot, and the frame rendered from the patched snapshot B's frame.  Arrays with bounding volumes
# move in all of them (obj_frametable.h:22,133, obj_aliencube.h:215, scn_demo03.h:171,177, the swarm's groups).
MOVED = [("demo02", [], 1, 2), ("demo03", [], 3, 4), ("demo01", [], 5, 6), ("test14", [], 3, 4), ("test16", [], 3, 4),
         ("demo03", ["--swarm", "96,5,1"], 7, 8)]


def run(scene, w, h, args, tmp):
    raw, qrs, tree = (os.path.join(tmp, n) for n in ("f.raw", "s.qrs", "t.json"))
    cmd = [REF, "--scene", scene, "-w", str(w), "-h", str(h), "--out", raw, "--snapshot", qrs, "--tree", tree] + args
    out = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(f"{scene}: {out.stdout}{out.stderr}")
    frame = np.fromfile(raw, dtype="<u4").reshape(h, w) & 0xFFFFFF
    return frame, open(qrs, "rb").read(), open(tree, "rb").read()


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = json.load(open(os.path.join(HERE, "manifest.json")))
    for name, m in sorted(manifest.items()):
        if m["w"] > 640:
            continue
        tmp = tempfile.mkdtemp(prefix="qrtree_")
        _, blob, tree = run(m["scene"], m["w"], m["h"], m["args"], tmp)
        committed = gzip.decompress(open(os.path.join(HERE, m["snapshot"]), "rb").read())
        if blob != committed:
            raise RuntimeError(f"{name}: the snapshot of this run differs from the committed one")
        with open(os.path.join(OUT, name + ".json.gz"), "wb") as f:
            f.write(gzip.compress(tree, 9, mtime=0))
        print(name, len(json.loads(tree)["nodes"]), "nodes")
    for name, (scene, args) in sorted(EXTRA.items()):
        tmp = tempfile.mkdtemp(prefix="qrtree_")
        frame, _, tree = run(scene, 160, 120, args, tmp)
        with open(os.path.join(OUT, name + ".json.gz"), "wb") as f:
            f.write(gzip.compress(tree, 9, mtime=0))
        bio = io.BytesIO(); np.save(bio, frame.astype("<u4"))
        with open(os.path.join(OUT, name + ".frame.npy.gz"), "wb") as f:
            f.write(gzip.compress(bio.getvalue(), 9, mtime=0))
        print(name, "frame + tree")


def jitter():
    for scene, seed in JITTER:
        tmp = tempfile.mkdtemp(prefix="qrtree_")
        frame, _, tree = run(scene, 160, 120, ["--jitter", str(seed), "--opts-off", "tiling"], tmp)
        name = "%s_160_jt%d" % (scene, seed)
        with open(os.path.join(OUT, name + ".json.gz"), "wb") as f:
            f.write(gzip.compress(tree, 9, mtime=0))
        bio = io.BytesIO(); np.save(bio, frame.astype("<u4"))
        with open(os.path.join(OUT, name + ".frame.npy.gz"), "wb") as f:
            f.write(gzip.compress(bio.getvalue(), 9, mtime=0))
        print(name, "frame + tree")


def fuzz():
    import struct
    for scene, seed, args in FUZZ:
        tmp = tempfile.mkdtemp(prefix="qrtree_")
        _, blob, tree = run(scene, 32, 24, (["--jitter", str(seed)] if seed else []) + args, tmp)
        b = bytearray(blob)
        n_texels, off_texels = struct.unpack_from("<I", b, 4 * 9)[0], struct.unpack_from("<I", b, 4 * 16)[0]
        b[off_texels:off_texels + 4 * n_texels] = bytes(4 * n_texels)
        name = "fuzz_%s_s%d%s" % (scene, seed, "_swarm" if "--swarm" in args else ("_noopt" if args else ""))
        with open(os.path.join(OUT, name + ".json.gz"), "wb") as f:
            f.write(gzip.compress(tree, 9, mtime=0))
        with open(os.path.join(OUT, name + ".qrs.gz"), "wb") as f:
            f.write(gzip.compress(bytes(b), 9, mtime=0))
        print(name)


def moved():
    import struct
    for scene, args, sa, sb in MOVED:
        for seed, keep_frame in ((sa, False), (sb, True)):
            tmp = tempfile.mkdtemp(prefix="qrtree_")
            frame, blob, tree = run(scene, 64, 48, ["--shift", str(seed)] + args, tmp)
            b = bytearray(blob)
            n_texels, off_texels = struct.unpack_from("<I", b, 4 * 9)[0], struct.unpack_from("<I", b, 4 * 16)[0]
            name = "moved_%s%s_s%d" % (scene, "_swarm" if args else "", seed)
            if keep_frame:
                bio = io.BytesIO(); np.save(bio, frame.astype("<u4"))
                with open(os.path.join(OUT, name + ".frame.npy.gz"), "wb") as f:
                    f.write(gzip.compress(bio.getvalue(), 9, mtime=0))
            else:
                # A keeps its texels: the patched snapshot is rendered (textures do not move)
                pass
            if keep_frame:
                b[off_texels:off_texels + 4 * n_texels] = bytes(4 * n_texels)
            with open(os.path.join(OUT, name + ".json.gz"), "wb") as f:
                f.write(gzip.compress(tree, 9, mtime=0))
            with open(os.path.join(OUT, name + ".qrs.gz"), "wb") as f:
                f.write(gzip.compress(bytes(b), 9, mtime=0))
            print(name)


if __name__ == "__main__":
    if sys.argv[1:] == ["fuzz"]:
        fuzz()
    elif sys.argv[1:] == ["moved"]:
        moved()
    elif sys.argv[1:] == ["jitter"]:
        jitter()
    else:
        main()
        jitter()
        fuzz()
        moved()
