#!/usr/bin/env python3
"""Walk statistics of one snapshot with the QR_STATS build: tools/gpu_stats.py NAME [depth] (GPU box).
Build first: make -C quadray-engine_amd/csrc variant NAME=stats EXTRA=-DQR_STATS"""
import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QR_LIB"] = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_stats.so")
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", sys.argv[1] + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
if len(sys.argv) > 2:
    scn.set_depth(int(sys.argv[2]))
print(sys.argv[1], "depth", scn.info.depth, flush=True)
scn.render_count()
