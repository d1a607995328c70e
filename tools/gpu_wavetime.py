#!/usr/bin/env python3
"""Per-wave timeline of one launch (QR_WAVETIME build): tools/gpu_wavetime.py NAME [depth] (GPU box).
Build first: make -C quadray-engine_amd/csrc variant NAME=wt EXTRA=-DQR_WAVETIME"""
import os, sys, gzip
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QR_LIB"] = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_wt.so")
out = os.path.join(ROOT, "gpurun_out", "wavetime_" + sys.argv[1].replace(":", "_") + ".bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["QR_WAVETIME_OUT"] = out
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
if sys.argv[1].startswith("synth:"):
    import bench
    blob = bench.load_blob(sys.argv[1])
else:
    blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", sys.argv[1] + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob, rebin_tiles=sys.argv[1].startswith("synth:"))
if len(sys.argv) > 2:
    scn.set_depth(int(sys.argv[2]))
f = scn.new_frame()
scn.render(f); torch.cuda.synchronize()
import time
t_w = time.time()
while time.time() - t_w < 2.0:                       # two seconds of back-to-back launches first: the clock the chip settles at
    for _ in range(50):
        scn.render(f)
    torch.cuda.synchronize()
avg, mn = scn.render_timed(f, 5)
w = np.fromfile(out, dtype=np.uint64).reshape(-1, 14)
w = w[w[:, 2] != 0]
t0 = int(w[:, 0].min())
st = (w[:, 0].astype(np.int64) - t0) / 100.0          # us
mid = (w[:, 1].astype(np.int64) - t0) / 100.0
en = (w[:, 2].astype(np.int64) - t0) / 100.0
dur = en - st
walks = (w[:, 3] >> np.uint64(40)).astype(np.int64)
print(f"{sys.argv[1]}: kernel {avg*1e3:.1f} us; waves {len(w)}; last wave ends at {en.max():.1f} us")
# in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles / 100 MHz ticks over each wave's life, waves of >= 5 us
cyc = (w[:, 13].astype(np.int64) - w[:, 12].astype(np.int64)); tk = (w[:, 2].astype(np.int64) - w[:, 0].astype(np.int64))
m = tk >= 500
if m.any():
    clk = cyc[m] / tk[m] * 100.0
    print("in-kernel clock MHz over waves >= 5 us: p10 %.0f p50 %.0f p90 %.0f (%d waves)" % (*np.percentile(clk, [10, 50, 90]), int(m.sum())))
print("whole launch: first stamp to last stamp %.1f us, %.0f MHz" % ((w[:, 2].max() - w[:, 0].min()) / 100.0,
      (int(w[:, 13].max()) - int(w[:, 12].min())) / max(1, int(w[:, 2].max()) - int(w[:, 0].min())) * 100.0))
print("wave duration us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(dur, [10, 50, 90, 99, 100])))
print("first traverse (start->after) us: p50 %.1f p90 %.1f" % tuple(np.percentile(mid - st, [50, 90])))
print("non-shadow walks per wave: mean %.2f max %d; waves with >1: %d" % (walks.mean(), walks.max(), (walks > 1).sum()))
for lo, hi in [(1, 1), (2, 3), (4, 8), (9, 1000)]:
    m = (walks >= lo) & (walks <= hi)
    if m.any():
        print(f"  walks {lo}-{hi}: {m.sum()} waves, duration mean {dur[m].mean():.1f} us max {dur[m].max():.1f}, start p50 {np.median(st[m]):.1f}, end max {en[m].max():.1f}")
# the slowest waves: how long are their dependent chains?
idx = np.argsort(-dur)[:8]
for i in idx:
    print(f"  slow wave: {dur[i]:7.1f} us start {st[i]:6.1f}  rounds {int(w[i,7])} nearest-hit groups {int(w[i,5])} shadow groups {int(w[i,6])}"
          f"; us in traverse {int(w[i,8])/100:.1f}, shade {int(w[i,9])/100:.1f} of which shadow walks {int(w[i,10])/100:.1f};"
          f" cells looked at {int(w[i,11]) & 0xFFFF} / shadow {(int(w[i,11]) >> 16) & 0xFFFF}, solved {(int(w[i,11]) >> 32) & 0xFFFF} / shadow {(int(w[i,11]) >> 48) & 0xFFFF}")
# where the slowest waves are: footprint coordinates (schedule entry), and what their pixels see (primary hit ids)
ids = torch.zeros((scn.height, scn.width), dtype=torch.int32, device="cuda")
scn.render(f, ids=ids); torch.cuda.synchronize()
idn = ids.cpu().numpy()
fw, fh = (8, 8) if scn.info.fsaa == 0 else ((8, 4) if scn.info.fsaa == 1 else (4, 4))
for i in idx:
    o = int(w[i, 4]); bx, by = o & 0x3FFF, (o >> 14) & 0x3FFF
    tile = idn[by * fh:(by + 1) * fh, bx * fw:(bx + 1) * fw]
    u, c = np.unique(tile, return_counts=True)
    print(f"  list groups walked: nearest-hit {int(w[i,5])} shadow {int(w[i,6])} in {int(w[i,7])} rounds;", end="")
    print(f"  slow wave at pixels x {bx*fw}..{bx*fw+fw-1} y {by*fh}..{by*fh+fh-1} heavy {o >> 30}: primary hits (surface*2+side: count) " + " ".join(f"{a}:{b}" for a, b in zip(u, c)))
tot_cells = w[:, 4].astype(np.int64) * 0 + w[:, 5].astype(np.int64)
print(f"all waves: cells per wave mean {tot_cells.mean():.1f}; sum of wave durations / sum of cells = {dur.sum()*1e3/max(1,tot_cells.sum()):.0f} ns per cell")
# occupancy over time: waves in flight sampled every 5% of the kernel
T = en.max()
for q in np.linspace(0.05, 1.0, 20):
    t = q * T
    print(f"  t={t:7.1f} us in flight {int(((st <= t) & (en > t)).sum())}", end="")
print()
