#!/usr/bin/env python3
"""Fixtures for the list-building pass (tests/test_lists.py; build container only).

qr_snapshot_build_lists_c rebuilds the per-side surface lists, light lists and per-side shadow lists of a scene from its
global list and the engine's box predicates (csrc/qr_sides.cpp).  The predicates walk the scene's HIERARCHY (an array seen
from one side of a surface hands that side to everything under it, engine.cpp:2222-2330), and the engine removes the
bounding-volume elements from a camera list when its screen tiling is on (engine.cpp:1711-1725) -- so the snapshots here are
captured with that one optimisation off (`qr_ref --opts-off tiling`): same scene, same per-surface lists, the camera list
keeps the hierarchy.  For every small case of tests/golden/manifest.json (except the `--opts none` ones, whose engine lists
are not the optimised ones) this stores (for the cases KEEP selects)
    lists/<case>.qrs.gz     the snapshot (it carries the engine's own lists: the expected output of the pass)
and checks that the reference frame is the tiled fixture's (same hash), so the frames of tests/golden/ serve both.
Data only: nothing of the reference's source text is stored.
"""
import gzip, json, os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
OUT = os.path.join(HERE, "lists")
# every swarm, every CORE test scene and every engine-jittered scene; of the demo scenes one or two times each (their other
# fixtures differ in camera, options and size, which the lists do not depend on)
KEEP = re.compile(r"^(swarm_.*|test\d\d_160|.*_j\d+|demo01_160|demo01_160_gf_t5000|demo02_160|demo02_160_gf_t5000|demo03_160|"
                  r"demo03_160_aa2_t2500|demo02_odd_33x17_aa4)$")


def main():
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="qrlists_"); os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)
    out = {}
    for name, e in sorted(man.items()):
        if "frame" not in e and not name.startswith("swarm_"):
            continue
        if "--opts" in e["args"]:
            continue
        if not KEEP.match(name):
            continue
        qrs = os.path.join(tmp, "s.qrs")
        cmd = [REF, "--scene", e["scene"], "-w", str(e["w"]), "-h", str(e["h"])] + e["args"] + ["--opts-off", "tiling", "--snapshot", qrs]
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(name + ": " + r.stdout + r.stderr)
        h = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("hash ")][0]
        if h != e["hash"]:
            raise RuntimeError(f"{name}: the frame without screen tiling differs from the tiled fixture's")
        blob = open(qrs, "rb").read()
        with open(os.path.join(OUT, name + ".qrs.gz"), "wb") as f:
            f.write(gzip.compress(blob, 9, mtime=0))
        out[name] = dict(snapshot=name + ".qrs.gz", snapshot_bytes=len(blob), hash=h)
        print(name, h, len(blob))
    json.dump(out, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
