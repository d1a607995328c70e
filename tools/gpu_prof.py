#!/usr/bin/env python3
"""Walk statistics of one workload with the QR_PROF build: tools/gpu_stats.py NAME|synth:N:W:H:D [depth] (GPU box).
Build first: make -C quadray-engine_amd/csrc variant NAME=prof EXTRA=-DQR_PROF"""
import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QR_LIB"] = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_prof.so")
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
name = sys.argv[1]
if name.startswith("synth:"):
    import bench
    blob = bench.load_blob(name)
else:
    blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", name + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob, rebin_tiles=name.startswith("synth:"))
if len(sys.argv) > 2:
    scn.set_depth(int(sys.argv[2]))
print(name, "depth", scn.info.depth, flush=True)
_, c = scn.render_count()
print(c.as_dict(), flush=True)
import numpy as np
f = scn.new_frame(); scn.render(f); torch.cuda.synchronize(); print("hash %016x" % qr.frame_hash(f), flush=True)
