#!/usr/bin/env python3
"""Path-tracer mode against the reference's frames (GPU box): tools/gpu_pt_probe.py SNAPSHOT.qrs REF_N.raw N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from qr_loader import load_package
qr = load_package()
import gzip
blob = gzip.decompress(open(sys.argv[1], "rb").read())
n = int(sys.argv[3])
scn = qr.Scene(blob)
ref = np.frombuffer(gzip.decompress(open(sys.argv[2], "rb").read()), dtype=np.uint32).reshape(scn.height, scn.width)
scn.set_pt(True)
f = scn.new_frame()
for _ in range(n):
    scn.render(f)
torch.cuda.synchronize()
out = f.cpu().numpy().view(np.uint32)
def rgb(a): return np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.float64)
g, r = rgb(out), rgb(ref)
print("frames", n, "gpu mean", g.mean((0, 1)).round(2), "ref mean", r.mean((0, 1)).round(2))
bh, bw = 8, 8
gb = g.reshape(scn.height // bh, bh, scn.width // bw, bw, 3).mean((1, 3)); rb = r.reshape(scn.height // bh, bh, scn.width // bw, bw, 3).mean((1, 3))
d = np.abs(gb - rb)
print("8x8 block means: mean abs diff %.2f, max %.2f (of 255)" % (d.mean(), d.max()))
np.save(os.path.join(ROOT, "gpurun_out", "pt_gpu_n%d.npy" % n), out)
