#!/usr/bin/env python3
"""Time the synthetic scene on the GPU: tools/gpu_synth_time.py N W H [depth] (GPU box)."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
spec = importlib.util.spec_from_file_location("qr_synth", os.path.join(ROOT, "quadray-engine_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
n, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
t = time.time(); blob = synth.make_scene(n_objects=n, width=w, height=h, depth=depth, leaf=int(os.environ.get('QR_SYNTH_LEAF', '4'))); t_gen = time.time() - t
t = time.time(); scn = qr.Scene(blob, rebin_tiles=True); t_up = time.time() - t
f = scn.new_frame()
t = time.time(); _, c = scn.render_count(f); t_cnt = time.time() - t
avg, mn = scn.render_timed(f, 3)
print(f"synth n={n} {w}x{h} depth {depth}: generate {t_gen:.2f}s upload+bin {t_up:.2f}s cells {scn.info.n_elm} tiles {scn.info.n_tiles} "
      f"count-kernel {t_cnt:.2f}s; kernel avg {avg:.2f} ms; rays {c.total()/1e6:.1f}M -> {c.total()/avg/1e3:.0f} Mrays/s", flush=True)
