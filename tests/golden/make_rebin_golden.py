#!/usr/bin/env python3
"""Fixtures that pin what GPU tile binning (QR_UPLOAD_REBIN_TILES / QR_REBIN=1) renders (build container only).

The drop-in fuzz of round 2 found jittered scenes whose frame with rebuilt tile lists differs from the engine's frame.
The UNMODIFIED reference shows why by itself: on those scenes its frame with screen tiling ON (RT_OPTS_TILING |
RT_OPTS_TILING_EXT1, engine.cpp:1956-2128, 3129-3253) differs from its own frame with ONLY that optimisation off
(oracle/_ref/qr_ref --opts-off tiling): its tiling drops surfaces from tiles they cover.  For every jittered scene of
the fuzz (tools/gpu_dropin_fuzz.sh: six scenes x seeds 1..6 at 200x150) the reference renders both frames; every scene
where they differ becomes a case here:
    <case>.qrs.gz          snapshot of the TILED frame (the engine's tile lists, the drop-in default)
    manifest.json          per case: reference args, hash of the tiled frame, hash of the tiling-off frame,
                           number of differing pixels
    <case>.tiled.npy.gz / <case>.untiled.npy.gz   both reference frames (the three smallest snapshots only)
tests/test_rebin_pin.py asserts: oracle / GPU with the engine's tile lists == the tiled frame; oracle with the camera
list in every tile and GPU with rebuilt tile lists == the tiling-off frame, bit for bit; and the two frames differ only in
pixels where the tiled frame has no primary hit.  Data only: nothing of the reference's source text is stored.
"""
import gzip, io, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
OUT = os.path.join(HERE, "rebin")
SCENES = ("test03", "test07", "test09", "test11", "test13", "test14")
SEEDS = range(1, 7)
W, H = 200, 150
KEEP_FRAMES = 3


def run(args, tmp):
    raw = os.path.join(tmp, "f.raw"); qrs = os.path.join(tmp, "s.qrs")
    out = subprocess.run([REF] + args + ["--out", raw, "--snapshot", qrs], cwd=tmp, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stdout + out.stderr)
    h = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("hash ")][0]
    return h, np.fromfile(raw, dtype="<u4").reshape(H, W) & 0xFFFFFF, open(qrs, "rb").read()


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="qrrebin_"); os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)
    manifest, frames, same = {}, {}, []
    for sc in SCENES:
        for seed in SEEDS:
            base = ["--scene", sc, "-w", str(W), "-h", str(H), "--jitter", str(seed)]
            h_t, f_t, blob = run(base, tmp)
            h_u, f_u, _ = run(base + ["--opts-off", "tiling"], tmp)
            name = f"{sc}_j{seed}"
            if h_t == h_u:
                same.append(name)
                continue
            with open(os.path.join(OUT, name + ".qrs.gz"), "wb") as f:
                f.write(gzip.compress(blob, 9, mtime=0))
            manifest[name] = dict(scene=sc, w=W, h=H, args=base[5:], hash_tiled=h_t, hash_untiled=h_u,
                                  differing_pixels=int((f_t != f_u).sum()), snapshot=name + ".qrs.gz", snapshot_bytes=len(blob))
            frames[name] = (f_t, f_u)
            print(name, h_t, h_u, manifest[name]["differing_pixels"], len(blob))
    for name in sorted(manifest, key=lambda n: manifest[n]["snapshot_bytes"])[:KEEP_FRAMES]:
        for tag, fr in zip(("tiled", "untiled"), frames[name]):
            bio = io.BytesIO(); np.save(bio, fr.astype("<u4"))
            with open(os.path.join(OUT, f"{name}.{tag}.npy.gz"), "wb") as f:
                f.write(gzip.compress(bio.getvalue(), 9, mtime=0))
            manifest[name]["frame_" + tag] = f"{name}.{tag}.npy.gz"
    json.dump(dict(cases=manifest, scenes_where_tiling_changes_nothing=same), open(os.path.join(OUT, "manifest.json"), "w"),
              indent=1, sort_keys=True)
    print(f"{len(manifest)} scenes where the reference's tiled and untiled frames differ, {len(same)} where they agree")


if __name__ == "__main__":
    main()
