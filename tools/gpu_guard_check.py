#!/usr/bin/env python3
"""The QR_STATS + QR_GUARD build (every cell offset of the per-lane walks checked before it is loaded, csrc/qr_walk.hpp
QR_GUARD_POS) on scenes that drive walk_div / walk_pool / walk_dda hard: synthetic crowds with built lists, the 10 000-object
scene at a small size, a swarm fixture.  Prints one line per case: "<case> guard_bad 0 frame_ok 1"; exit code 1 on any bad
offset or frame.  Build: make -C quadray-engine_amd/csrc guard.  (GPU box; run by tests/test_synth.py in a child process.)"""
import gzip, importlib.util, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "quadray-engine_amd", "libqrhip_guard.so")
if os.environ.get("QR_LIB") != LIB:
    # the library is chosen at import time: run ourselves again with it selected, collect stderr (QR_GUARD lines)
    env = dict(os.environ, QR_LIB=LIB)
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    bad = [l for l in p.stderr.splitlines() if l.startswith("QR_GUARD")]
    sys.stdout.write(p.stdout)
    for l in bad:
        print(l)
    if p.returncode != 0:
        print("child failed:\n" + p.stderr[-3000:])
    sys.exit(1 if (p.returncode != 0 or bad) else 0)

sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from qr_loader import load_package
import qr_oracle
qr = load_package()
spec = importlib.util.spec_from_file_location("qr_synth", os.path.join(ROOT, "quadray-engine_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)

cases = {
    "synth2000_built": qr.build_lists(synth.make_scene(shadow_lists=False, n_objects=2000, width=384, height=216, depth=4, box=45.0, gamma=True, fsaa=2)),
    "synth10k_built_640": qr.build_lists(synth.make_scene(shadow_lists=False, n_objects=10000, width=640, height=360, depth=4)),
    "synth300_own_lists": synth.make_scene(n_objects=300, width=320, height=240, depth=4, box=20.0),
    "swarm_demo01_240_mix": gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "swarm_demo01_240_mix.qrs.gz"), "rb").read()),
}
# a dense cloud with lowered grid thresholds (every list of 64 members gets its uniform grid, the plane its shadow grids): an image of
# ~300 MB, i.e. valid cell offsets beyond the 256 MB bound the guard had until round 3
cases["dense490_low_thresholds"] = qr.build_lists(synth.make_scene(shadow_lists=False, n_objects=490, width=320, height=180, depth=6, box=5.96, seed=20076))
rc = 0
for name, blob in cases.items():
    low = name.endswith("low_thresholds")
    if low:
        os.environ.update({"QR_DDA": "64", "QR_GRID": "64"})
    try:
        scn = qr.Scene(blob, rebin_tiles=name.startswith("synth"))
    finally:
        if low:
            del os.environ["QR_DDA"], os.environ["QR_GRID"]
    if low:
        print(f"{name} image bytes {scn.info.device_bytes}", flush=True)
    frame, counts = scn.render_count()          # the counting launch prints "QR_GUARD n bad cell offsets" on stderr when n != 0
    torch.cuda.synchronize()
    out = frame.cpu().numpy().view(np.uint32)
    ref, _, _ = qr_oracle.render(blob, threads=16)
    ok = bool((out == ref).all())
    print(f"{name} frame_ok {int(ok)} rays {counts.as_dict()}", flush=True)
    if not ok:
        rc = 1
sys.exit(rc)
