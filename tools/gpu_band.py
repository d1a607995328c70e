#!/usr/bin/env python3
"""Render one band of rows alone, a few times (for rocprofv3 --pmc runs on the slowest footprints of a frame):
tools/gpu_band.py NAME y0 y1 [iters]  (GPU box)."""
import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", sys.argv[1] + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
f = scn.new_frame()
scn.set_rows(int(sys.argv[2]), int(sys.argv[3]), 0, 1)
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 6):
    scn.render(f); torch.cuda.synchronize()
avg, mn = scn.render_timed(f, 5)
print(f"band {sys.argv[2]}..{sys.argv[3]}: avg {avg*1e3:.1f} us min {mn*1e3:.1f} us")
