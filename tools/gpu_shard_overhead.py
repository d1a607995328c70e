#!/usr/bin/env python3
"""Compute-side cost of the multi-GPU schedule on ONE GPU: per step, the N row blocks rank 0 would render
(one block of each of N frames = one frame of work).  tools/gpu_shard_overhead.py [workload snapshot] (GPU box)."""
import os, sys, gzip, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
spec = importlib.util.spec_from_file_location("qr_sharding", os.path.join(ROOT, "quadray-engine_amd", "sharding.py"))
sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
name = sys.argv[1] if len(sys.argv) > 1 else "c2b_demo01_1080p"
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", name + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
H, W = scn.height, scn.width
for N in (1, 2, 4, 8):
    ex = sh.FrameExchange(H, W, N, 0)
    frames = [scn.new_frame() for _ in range(N)]
    streams = [torch.cuda.Stream() for _ in range(N)]
    def step():
        for f in range(N):
            r0, r1 = ex.my_rows(f)
            scn.set_rows(r0, r1, 0, 1)
            scn.render(frames[f], stream=streams[f])
    for _ in range(20): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    K = 200
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K * 1e3
    print(f"{name}: N={N}: {dt:.4f} ms per step (one frame of work in {N} launches)", flush=True)
