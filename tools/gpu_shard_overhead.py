#!/usr/bin/env python3
"""Compute-side cost of the multi-GPU schedule on ONE GPU: per step, the N row blocks rank 0 would render
(one block of each of N frames = one frame of work), D steps in flight.
mode A: every (step slot, block) has its own stream; mode B: one stream per step slot, blocks back to back; mode C: one multi-target launch per step.
tools/gpu_shard_overhead.py [snapshot] (GPU box)."""
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
D = 3
for mode in "ABC":
    for N in (1, 2, 4, 8):
        ex = sh.FrameExchange(H, W, N, 0)
        frames = [[scn.new_frame() for _ in range(N)] for _ in range(D)]
        streams = [[torch.cuda.Stream() for _ in range(N if mode == "A" else 1)] for _ in range(D)]
        multi = [qr.MultiRender([(scn, frames[b][f]) + ex.my_rows(f) for f in range(N)]) for b in range(D)]
        def step(i):
            b = i % D
            if mode == "C":                         # one launch for the N blocks
                multi[b](stream=streams[b][0])
                return
            for f in range(N):
                r0, r1 = ex.my_rows(f)
                scn.set_rows(r0, r1, 0, 1)
                scn.render(frames[b][f], stream=streams[b][f if mode == "A" else 0])
        for i in range(30): step(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        K = 600
        for i in range(K): step(i)
        t_issue = (time.perf_counter() - t) / K * 1e3
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / K * 1e3
        print(f"{name}: mode {mode} N={N}: {dt:.4f} ms per step, host issue time {t_issue:.4f} ms per step", flush=True)
