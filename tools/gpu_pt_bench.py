#!/usr/bin/env python3
"""Path-tracer frame times on the GPU box: tools/gpu_pt_bench.py [SNAPSHOT.qrs.gz] [frames]
Both modes (fast statistical kernel, eager machine), HIP events around `frames` launches after a warm-up."""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "pt", "test18_1080p_pt.qrs.gz")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scn = qr.Scene(gzip.decompress(open(path, "rb").read()))
f = scn.new_frame()
for eager in (False, True):
    scn.set_pt(True, eager=eager)
    for _ in range(3):
        scn.render(f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        scn.render(f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%dx%d depth %d %s: %.3f ms per frame (one sample per pixel), %.1f Msamples/s" %
          (scn.width, scn.height, scn.info.depth, "eager" if eager else "fast", ms, scn.width * scn.height / ms / 1e3))
