#!/usr/bin/env python3
"""Timeline of the DEFAULT bench mode (three launches in flight, one HIP stream each) without the profiler (GPU box).

  tools/gpu_timeline.py events [N] [WORKLOAD]   product library: a start and a stop HIP event around every launch on its stream
  tools/gpu_timeline.py waves  [N] [WORKLOAD]   QR_WAVETIME build (make variant NAME=wt EXTRA=-DQR_WAVETIME): every wave stamps
                                                s_memrealtime (100 MHz, one counter for the whole chip) at its start and end; N
                                                launches of N uploads of the same scene (a scene holds the stamps of its last launch)
Prints, and writes to gpurun_out/timeline_<mode>.txt: wall time per frame, the share of time with 1 / 2 / 3 launches resident
(first to last instruction of a launch), and for `waves` the wave slots occupied over time.
"""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "events"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
wl = sys.argv[3] if len(sys.argv) > 3 else "demo1_1080p"
if mode == "waves":
    os.environ["QR_LIB"] = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_wt.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from qr_loader import load_package
qr = load_package()
snap = bench.WORKLOADS[wl][0]
blob = bench.load_blob(snap)
D = 3
streams = [torch.cuda.Stream() for _ in range(D)]
lines = []


def say(s=""):
    print(s, flush=True); lines.append(s)


def residency(iv, skip):
    """iv: [(start_us, end_us)] per launch in launch order; statistics over launches skip..n-skip"""
    iv = np.asarray(iv, dtype=np.float64)
    mid = iv[skip:len(iv) - skip]
    t0, t1 = mid[0, 0], mid[-1, 0]                      # from the start of the first counted launch to the start of the last
    per_frame = (t1 - t0) / (len(mid) - 1)
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    cur, tp, hist = 0, ev[0][0], {}
    for t, d in ev:
        lo, hi = max(tp, t0), min(t, t1)
        if hi > lo:
            hist[cur] = hist.get(cur, 0.0) + (hi - lo)
        tp = t; cur += d
    tot = sum(hist.values())
    return per_frame, {k: v / tot for k, v in sorted(hist.items())}, float((mid[:, 1] - mid[:, 0]).mean())


say(f"# timeline of {n} launches of {wl}, {D} in flight on {D} streams, mode {mode}, lib {qr.lib().qr_version().decode()}")
if mode == "events":
    scn = qr.Scene(blob, rebin_tiles=snap.startswith("synth:"))
    frames = [scn.new_frame() for _ in range(D)]
    for rep in range(3):
        for i in range(30):
            scn.render(frames[i % D], stream=streams[i % D])
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    base.record(streams[0])
    for i in range(n):
        st = streams[i % D]
        e0[i].record(st)
        scn.render(frames[i % D], stream=st)
        e1[i].record(st)
    torch.cuda.synchronize()
    iv = [(base.elapsed_time(e0[i]) * 1e3, base.elapsed_time(e1[i]) * 1e3) for i in range(n)]
    say("# per launch: [start event, stop event] on its stream, us since the first record (the start event of launch i completes when")
    say("# launch i-3 on the same stream has ended, so the interval is an upper bound of the launch's residency)")
else:
    scns = [qr.Scene(blob, rebin_tiles=snap.startswith("synth:")) for _ in range(n)]
    frames = [scns[0].new_frame() for _ in range(D)]
    L = qr.lib()
    L.qr_wavetime_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    for rep in range(3):
        for i in range(n):
            scns[i].render(frames[i % D], stream=streams[i % D])
    torch.cuda.synchronize()
    buf = np.zeros(16 * 1024 * 1024, dtype=np.uint64)
    for s in scns:                                       # clear the warm-up's stamps
        L.qr_wavetime_read(s._h, buf.ctypes.data_as(ctypes.c_void_p), 0, 1)
    for i in range(n):
        scns[i].render(frames[i % D], stream=streams[i % D])
    torch.cuda.synchronize()
    waves = []
    for i, s in enumerate(scns):
        slots = L.qr_wavetime_read(s._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, 0)
        w = buf[: (buf.size // slots) * slots].reshape(-1, slots)
        w = w[w[:, 2] != 0][:, :3].astype(np.int64).copy()
        waves.append(w)
    t_base = min(int(w[:, 0].min()) for w in waves)
    iv = [((int(w[:, 0].min()) - t_base) / 100.0, (int(w[:, 2].max()) - t_base) / 100.0) for w in waves]
    say("# per launch: [first wave's first instruction, last wave's last instruction], us (s_memrealtime stamps, 100 MHz)")

skip = max(3, n // 8)
per_frame, share, mean_res = residency(iv, skip)
say(f"launches counted: {n - 2 * skip} (the first and last {skip} of {n} left out: ramp and drain)")
say(f"wall time per frame: {per_frame:.2f} us   (mean residency of one launch {mean_res:.1f} us)")
say("share of time with k launches resident: " + ", ".join(f"{k}: {v:.3f}" for k, v in share.items()))
say(f"time with >= 2 launches resident: {sum(v for k, v in share.items() if k >= 2):.3f}")
for i in range(skip, min(n - skip, skip + 12)):
    say(f"  launch {i:3d} stream {i % D}: {iv[i][0]:9.1f} .. {iv[i][1]:9.1f} us  ({iv[i][1] - iv[i][0]:6.1f})")
if mode == "waves":
    st = np.concatenate([(w[:, 0] - t_base) / 100.0 for w in waves]); en = np.concatenate([(w[:, 2] - t_base) / 100.0 for w in waves])
    t0, t1 = iv[skip][0], iv[n - skip][0]
    ts = np.arange(t0, t1, 1.0)
    tt = np.concatenate([st, en]); dd = np.concatenate([np.ones_like(st), -np.ones_like(en)])
    o = np.argsort(tt, kind="stable"); tt = tt[o]; cum = np.cumsum(dd[o])
    occ = cum[np.clip(np.searchsorted(tt, ts, side="right") - 1, 0, len(cum) - 1)]
    say(f"wave slots occupied, sampled every us between launch {skip} and launch {n - skip}: mean {occ.mean():.0f} p10 {np.percentile(occ, 10):.0f} "
        f"p50 {np.percentile(occ, 50):.0f} p90 {np.percentile(occ, 90):.0f} max {occ.max()} of 4096 (256 CUs x 4 SIMDs x 4 waves)")
    dur = en - st
    say(f"waves {len(dur)}: sum of wave durations {dur.sum() / 1e6:.3f} s; per frame {dur.sum() / n:.0f} wave-us = {dur.sum() / n / 4096:.1f} us of 4096 slots")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", f"timeline_{mode}_{wl}.txt"), "w").write("\n".join(lines) + "\n")
