#!/usr/bin/env python3
"""Phases of an upload (QR_VERBOSE=2: validate, bounds, rebin, compile) of one workload: tools/gpu_upload_time.py WORKLOAD (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QR_VERBOSE"] = "2"
import bench
from qr_loader import load_package
qr = load_package()
snap = bench.WORKLOADS[sys.argv[1]][0]
t = time.time(); blob = bench.load_blob(snap); print("snapshot (+ list building for synthetic scenes) %.1f ms" % ((time.time() - t) * 1e3), flush=True)
for k in range(2):
    t = time.time(); scn = qr.Scene(blob, rebin_tiles=snap.startswith("synth:")); print("upload %d: %.1f ms" % (k, (time.time() - t) * 1e3), flush=True)
