#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

For every case below the UNMODIFIED reference (oracle/_ref/qr_ref_shim, built from /root/reference by
oracle/Makefile) renders the frame with its own CPU SIMD backend and, through the drop-in shim, captures the
flattened scene snapshot.  Fixtures are data only:
    <case>.qrs.gz      snapshot (include/qr_scene.h layout), gzip
    <case>.frame.npy.gz  reference frame, uint32 0x00RRGGBB (small cases only)
    manifest.json      per case: reference args, FNV-1a-64 frame hash, sizes
Nothing of the reference's source text is stored.
"""
import gzip, io, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")

def fnv1a64(frame):
    h = 0xcbf29ce484222325
    data = (frame.astype(np.uint32) & 0xFFFFFF).astype("<u4").tobytes()
    # vectorised FNV is awkward; frames are <= 8M pixels, do it in chunks with python ints
    for b in data:
        h ^= b
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h

CASES = []
def case(name, scene, w, h, args=(), keep_frame=True):
    CASES.append(dict(name=name, scene=scene, w=w, h=h, args=list(args), keep_frame=keep_frame))

for d in ("demo01", "demo02", "demo03"):
    case(f"{d}_160", d, 160, 120)
    case(f"{d}_160_gf_t5000", d, 160, 120, ["--gamma", "--fresnel", "-t", "5000"])
    case(f"{d}_160_gf_aa4", d, 160, 120, ["--gamma", "--fresnel", "--fsaa", "4"])
    case(f"{d}_160_aa2_t2500", d, 160, 120, ["--fsaa", "2", "-t", "2500"])
for i in range(1, 19):
    case(f"test{i:02d}_160", f"test{i:02d}", 160, 120)
for i in (5, 12, 14, 16):
    case(f"test{i:02d}_160_noopt", f"test{i:02d}", 160, 120, ["--opts", "none"])
case("demo01_160_d0", "demo01", 160, 120, ["--depth", "0"])
case("demo02_160_gf_d3", "demo02", 160, 120, ["--gamma", "--fresnel", "--depth", "3"])
case("demo01_odd_157x93", "demo01", 157, 93)            # ragged: width not a multiple of any tile / SIMD width
case("demo02_odd_33x17_aa4", "demo02", 33, 17, ["--fsaa", "4", "--gamma"])
# other cameras of the scenes (rt_Scene::next_cam), other animation times, recursion depths, option mixes
for d in ("demo01", "demo02", "demo03"):
    if d != "demo01":                                   # demo01 has a single camera
        case(f"{d}_160_cam1", d, 160, 120, ["--camera", "1"])
    case(f"{d}_160_cam2_gf_aa2", d, 160, 120, ["--camera", "2", "--gamma", "--fresnel", "--fsaa", "2"])
case("demo01_160_t12345", "demo01", 160, 120, ["-t", "12345"])
case("demo02_160_t7777_gf", "demo02", 160, 120, ["-t", "7777", "--gamma", "--fresnel"])
case("demo03_160_t9999_aa4", "demo03", 160, 120, ["-t", "9999", "--fsaa", "4"])
case("demo02_160_d1", "demo02", 160, 120, ["--depth", "1"])
case("demo02_160_gf_d5", "demo02", 160, 120, ["--gamma", "--fresnel", "--depth", "5"])
case("test07_160_gf", "test07", 160, 120, ["--gamma", "--fresnel"])
case("test13_160_gf_aa4", "test13", 160, 120, ["--gamma", "--fresnel", "--fsaa", "4"])
case("test18_160_gf_t4000", "test18", 160, 120, ["--gamma", "--fresnel", "-t", "4000"])
case("demo03_320x240_aa4_gf", "demo03", 320, 240, ["--fsaa", "4", "--gamma", "--fresnel"])
# transforms fuzzed inside the engine (oracle/ref_driver.cpp --jitter SEED: every object of the scene's static description
# gets right-angle or arbitrary rotations, negative / non-unit scalers, a shifted position): scalers before and after
# rotations, nested transform nodes, mirrored axis maps -- against the reference itself, not only against the oracle
for scene, seed, extra in (("test05", 1, []), ("test06", 2, []), ("test07", 3, ["--gamma", "--fresnel"]), ("test08", 4, []),
                           ("test03", 18, []), ("test10", 6, ["--fsaa", "4"]), ("test11", 7, []), ("test12", 8, ["--opts", "none"]),
                           ("test15", 9, []), ("test17", 18, []), ("test09", 11, ["--opts", "none"]), ("test05", 14, ["--fsaa", "2"]),
                           ("test18", 18, ["--gamma", "--fresnel"])):
    case(f"{scene}_160_j{seed}", scene, 160, 120, ["--jitter", str(seed)] + extra)
# a crowd of small quadrics built INSIDE the engine (oracle/ref_driver.cpp --swarm N,SEED: arrays of 12 spheres with
# bounding volumes, every fourth group a transform node, bowls, ellipsoids, plain / metal / glass): the scene class of
# BASELINE config 5 with the engine's own lists, as far as its N^2 per-surface lists allow
case("swarm_demo01_240", "demo01", 160, 120, ["--swarm", "240,1"])
case("swarm_demo01_240_gf_aa4", "demo01", 160, 120, ["--swarm", "240,1", "--gamma", "--fresnel", "--fsaa", "4"])
case("swarm_demo03_200_t3000", "demo03", 160, 120, ["--swarm", "200,2", "-t", "3000"])
case("swarm_demo01_240_1080p", "demo01", 1920, 1080, ["--swarm", "240,1"], keep_frame=False)
# the same with every kind of quadric (--swarm N,SEED,1: cut cylinders, cones, paraboloids, hyperboloids, axis maps from
# right-angle turns); 24 more such scenes (3 stock scenes x 8 seeds) agreed with the oracle when these were made
case("swarm_demo01_240_mix", "demo01", 160, 120, ["--swarm", "240,3,1"])
case("swarm_demo02_200_mix_gf", "demo02", 160, 120, ["--swarm", "200,4,1", "--gamma", "--fresnel"])
# BASELINE.json configs
case("c1_demo01_640x480", "demo01", 640, 480)
case("c2_demo01_1080p_d0", "demo01", 1920, 1080, ["--depth", "0"], keep_frame=False)
case("c2b_demo01_1080p", "demo01", 1920, 1080, [], keep_frame=False)
case("c3_demo02_1080p_gf_d3", "demo02", 1920, 1080, ["--gamma", "--fresnel", "--depth", "3"], keep_frame=False)
case("c4_demo02_2160p_aa4_gf", "demo02", 3840, 2160, ["--gamma", "--fresnel", "--fsaa", "4"], keep_frame=False)

def main():
    only = set(sys.argv[1:])
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path)) if os.path.exists(man_path) else {}
    for c in CASES:
        if only and c["name"] not in only:
            continue
        tmp = tempfile.mkdtemp(prefix="qrgold_")
        os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)
        raw = os.path.join(tmp, "f.raw"); qrs = os.path.join(tmp, "s.qrs")
        cmd = [REF, "--scene", c["scene"], "-w", str(c["w"]), "-h", str(c["h"]), "--out", raw, "--snapshot", qrs] + c["args"]
        out = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError(f"{c['name']}: {out.stdout}{out.stderr}")
        ref_hash = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("hash ")][0]
        frame = np.fromfile(raw, dtype="<u4").reshape(c["h"], c["w"]) & 0xFFFFFF
        blob = open(qrs, "rb").read()
        with open(os.path.join(HERE, c["name"] + ".qrs.gz"), "wb") as f:
            f.write(gzip.compress(blob, 9, mtime=0))
        entry = dict(scene=c["scene"], w=c["w"], h=c["h"], args=c["args"], hash=ref_hash,
                     snapshot=c["name"] + ".qrs.gz", snapshot_bytes=len(blob))
        if c["keep_frame"]:
            bio = io.BytesIO(); np.save(bio, frame.astype("<u4"))
            with open(os.path.join(HERE, c["name"] + ".frame.npy.gz"), "wb") as f:
                f.write(gzip.compress(bio.getvalue(), 9, mtime=0))
            entry["frame"] = c["name"] + ".frame.npy.gz"
        manifest[c["name"]] = entry
        print(c["name"], ref_hash, len(blob))
    json.dump(manifest, open(man_path, "w"), indent=1, sort_keys=True)

if __name__ == "__main__":
    main()
