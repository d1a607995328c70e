#!/usr/bin/env python3
"""How long do the slowest footprints of a frame take when they have the chip to themselves?  Renders bands of 8 rows alone
(qr_scene_set_rows) and prints the launch time of each band beside the whole frame's: tools/gpu_lone.py NAME y0 y1 ... (GPU box)."""
import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", sys.argv[1] + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
f = scn.new_frame()
for _ in range(200):
    scn.render(f)
torch.cuda.synchronize()
avg, mn = scn.render_timed(f, 50)
print(f"whole frame: avg {avg*1e3:.1f} us min {mn*1e3:.1f} us")
H = scn.height
res = []
for y0 in (range(0, H, 8) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2:]]):
    scn.set_rows(y0, min(H, y0 + 8), 0, 1)
    scn.render(f); torch.cuda.synchronize()
    avg, mn = scn.render_timed(f, 20)
    res.append((mn * 1e3, y0))
res.sort(reverse=True)
print("bands of 8 rows rendered alone, slowest first (min of 20, us):", " ".join(f"y{y}:{t:.1f}" for t, y in res[:16]))
