#!/usr/bin/env python3
"""Kernel time of the synthetic 10k scene by recursion depth (GPU box): tools/gpu_synth_probe.py [synth:N:W:H:D]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "synth:10000:7680:4320:4"
blob = bench.load_blob(name)
scn = qr.Scene(blob, rebin_tiles=True)
f = scn.new_frame()
for d in (0, 1, 2, 4):
    scn.set_depth(d)
    scn.render(f); torch.cuda.synchronize()
    avg, mn = scn.render_timed(f, 3)
    print("depth %d: %.2f ms (min %.2f)  hash %016x" % (d, avg, mn, qr.frame_hash(f)), flush=True)
