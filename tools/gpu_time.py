#!/usr/bin/env python3
"""Time the render kernel on any golden snapshot: tools/gpu_time.py NAME [depth] (GPU box)."""
import sys, os, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
name = sys.argv[1]
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", name + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
if len(sys.argv) > 2:
    scn.set_depth(int(sys.argv[2]))
f = scn.new_frame()
scn.render(f); torch.cuda.synchronize()
avg, mn = scn.render_timed(f, 50)
print(f"{name} depth {scn.info.depth} {scn.width}x{scn.height}: avg {avg*1e3:.1f} us  min {mn*1e3:.1f} us  QR_DBG={os.environ.get('QR_DBG','0')}")
