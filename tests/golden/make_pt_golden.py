#!/usr/bin/env python3
"""Path-tracer fixtures (tests/golden/pt/), made in the build container from the unmodified reference (oracle/_ref):
snapshots captured in path-tracer mode through the shim, and frames of the reference's own path tracer after N
accumulated frames (qr_ref --pt N).  Scene: test18, the smallpt Cornell box (test/scenes/scn_test18.h), the one the
reference names for this mode (root/RooT.h:17-20); the demo scenes have point lights only and stay black.
usage: python tests/golden/make_pt_golden.py"""
import gzip, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF, SHIM = os.path.join(ROOT, "oracle", "_ref", "qr_ref"), os.path.join(ROOT, "oracle", "_ref", "qr_ref_shim")
OUT = os.path.join(ROOT, "tests", "golden", "pt")
CASES = [("test18_160_pt", [], [64, 512]), ("test18_160_gf_aa4_pt", ["--fsaa", "4", "--gamma", "--fresnel"], [128])]
os.makedirs(OUT, exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    for name, opts, counts in CASES:
        base = ["--scene", "test18", "-w", "160", "-h", "120"] + opts
        snap = os.path.join(tmp, name + ".qrs")
        subprocess.run([SHIM] + base + ["--pt", "1", "--snapshot", snap], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        with open(snap, "rb") as f, open(os.path.join(OUT, name + ".qrs.gz"), "wb") as g:
            g.write(gzip.compress(f.read(), 9, mtime=0))
        for n in counts:
            raw = os.path.join(tmp, "f.raw")
            subprocess.run([REF] + base + ["--pt", str(n), "--threads", "8", "--out", raw], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
            with open(raw, "rb") as f, open(os.path.join(OUT, f"{name}_n{n}.raw.gz"), "wb") as g:
                g.write(gzip.compress(f.read(), 9, mtime=0))
            print(name, n)
    # first frame at recursion depth 0: bit-exact pin of seeding, generator, jitter, emission, accumulation
    for name, opts in (("test18_160_pt_d0_n1", []), ("test18_160_aa4_pt_d0_n1", ["--fsaa", "4"])):
        raw = os.path.join(tmp, "f.raw")
        subprocess.run([REF, "--scene", "test18", "-w", "160", "-h", "120", "--depth", "0", "--pt", "1", "--out", raw] + opts,
                       check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        with open(raw, "rb") as f, open(os.path.join(OUT, name + ".raw.gz"), "wb") as g:
            g.write(gzip.compress(f.read(), 9, mtime=0))
        print(name)
    # the reference's first TWO frames at recursion depths 2 and 8, and its first frame at the default depth 10: the eager
    # path tracer (Scene.set_pt(True, eager=True)) follows the reference's shading order and must reproduce them
    for d, n in ((2, 2), (6, 2), (8, 1), (8, 2), (10, 1), (10, 2), (10, 4)):
        raw = os.path.join(tmp, "f.raw")
        subprocess.run([REF, "--scene", "test18", "-w", "160", "-h", "120", "--depth", str(d), "--pt", str(n), "--out", raw],
                       check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        with open(raw, "rb") as f, open(os.path.join(OUT, f"test18_160_pt_d{d}_n{n}.raw.gz"), "wb") as g:
            g.write(gzip.compress(f.read(), 9, mtime=0))
        print("eager", d, n)
    # the same with 4x FSAA, Gamma and Fresnel (the split of Fresnel surfaces draws numbers too): first three frames
    raw = os.path.join(tmp, "f.raw")
    subprocess.run([REF, "--scene", "test18", "-w", "160", "-h", "120", "--fsaa", "4", "--gamma", "--fresnel", "--pt", "3", "--out", raw],
                   check=True, cwd=tmp, stdout=subprocess.DEVNULL)
    with open(raw, "rb") as f, open(os.path.join(OUT, "test18_160_gf_aa4_pt_n3.raw.gz"), "wb") as g:
        g.write(gzip.compress(f.read(), 9, mtime=0))
