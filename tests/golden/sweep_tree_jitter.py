#!/usr/bin/env python3
"""Sweep for qr_hierarchy_apply (run in the build container only; needs oracle/_ref/qr_ref_shim).

For every scene and seed the UNMODIFIED reference draws every transform of the scene anew (oracle/ref_driver.cpp --jitter SEED)
and renders it twice: with its default options and with its screen tiling off (the tiling is not conservative on such
transforms, tests/test_rebin_pin.py).  The committed t = 0 snapshot of the scene is patched with the jittered tree
(qr_hierarchy_apply, QR_HIER_RESET_TILES | QR_HIER_BOUNDS), its lists are rebuilt (qr_snapshot_build_lists_c) and the oracle
renders it.  One line per case: differing pixels against both frames, the number of nodes whose transform node changed; when
pixels differ from the tiling-off frame, also the engine's OWN snapshot of the jittered scene with the camera list in every
tile and its per-surface lists rebuilt by the same pass -- if that frame equals ours, the difference is the engine's
bounding-volume arrays in its per-surface lists (they cut members off on these transforms), not the patched hierarchy.
usage: sweep_tree_jitter.py [--gpu] SEED [SEED ...] > profiles/rNN_hierarchy_jitter_sweep.txt
--gpu (on a GPU box; the reference runs on its host cores): the patched snapshot is ALSO uploaded with its tile lists rebuilt by
the GPU binning pass (QR_UPLOAD_REBIN_TILES) and rendered by the HIP kernel; its frame is compared with the oracle's.
"""
import json, os, struct, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_hierarchy as th
from conftest import load_blob
from qr_loader import load_package
import qr_oracle

REF = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
SCENES = ["demo01", "demo02", "demo03"] + ["test%02d" % k for k in range(1, 19)]


def run(scene, args, want_snapshot=False):
    tmp = tempfile.mkdtemp(prefix="qrsweep_")
    raw, qrs, tree = (os.path.join(tmp, n) for n in ("f.raw", "s.qrs", "t.json"))
    out = subprocess.run([REF, "--scene", scene, "-w", "160", "-h", "120", "--out", raw, "--snapshot", qrs, "--tree", tree] + args,
                         cwd=tmp, capture_output=True, text=True)
    if out.returncode:
        raise RuntimeError(out.stdout + out.stderr)
    frame = np.fromfile(raw, dtype="<u4").reshape(120, 160) & 0xFFFFFF
    return frame, json.load(open(tree)), (open(qrs, "rb").read() if want_snapshot else None)


def main():
    qr = load_package()
    gpu = "--gpu" in sys.argv[1:]
    seeds = [a for a in sys.argv[1:] if a != "--gpu"] or ["1", "2", "3"]
    n = same = refused = gpu_same = 0
    for scene in SCENES:
        base_name = scene + "_160"
        tb, base = th.load_tree(qr, base_name)
        s0 = qr.hierarchy_update(base, tb["opts"])
        for seed in seeds:
            tiled, t, snap = run(scene, ["--jitter", seed], True)
            untiled, _, _ = run(scene, ["--jitter", seed, "--opts-off", "tiling"])
            tgt = th.nodes_from_tree(qr, t)
            nxt = base.copy()
            for f in ("scl", "rot", "pos"):
                nxt[f] = tgt[f]
            s1 = qr.hierarchy_update(nxt, tb["opts"])
            changed = sum(1 for i in range(len(base)) if int(s0[i]["trnode"]) != int(s1[i]["trnode"]))
            n += 1
            try:
                blob = qr.hierarchy_apply(load_blob(base_name), nxt, tb["opts"], camera=tb["camera"], base=base,
                                          flags=qr.HIER_RESET_TILES | qr.HIER_BOUNDS)
            except qr.QrError as e:
                refused += 1
                print("%s seed %s: REFUSED %s (transform-node changes %d)" % (scene, seed, e, changed))
                continue
            built = qr.build_lists(blob)
            ours, _, _ = qr_oracle.render(built, threads=8)
            d_t, d_u = int((ours != tiled).sum()), int((ours != untiled).sum())
            note = ""
            if gpu:
                sc = qr.Scene(built, rebin_tiles=True)
                hip = sc.render().cpu().numpy().view(np.uint32) & 0xFFFFFF
                d_g = int((hip != ours).sum())
                gpu_same += d_g == 0
                note = "; HIP kernel with rebinned tiles: %d pixels from the oracle" % d_g
                del sc
            if d_u:
                h = struct.unpack_from("<4I6I7I5I", snap, 0)
                e2 = bytearray(snap)
                clist = struct.unpack_from("<i", snap, h[10] + 4 * 38)[0]
                for k in range(h[8]):
                    struct.pack_into("<i", e2, h[15] + 4 * k, clist)
                own, _, _ = qr_oracle.render(qr.build_lists(bytes(e2)), threads=8)
                note += "; the engine's own snapshot with lists rebuilt: %d pixels from ours" % int((own != ours).sum())
            else:
                same += 1
            print("%s seed %s: %d pixels from the tiled frame, %d from the tiling-off frame, transform-node changes %d%s"
                  % (scene, seed, d_t, d_u, changed, note))
    print("# %d cases, %d refused, %d equal to the reference's tiling-off frame pixel for pixel" % (n, refused, same)
          + (", HIP kernel equal to the oracle in %d" % gpu_same if gpu else ""))


if __name__ == "__main__":
    main()
