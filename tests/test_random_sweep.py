"""Random synthetic scenes through the GPU path and the oracle.  Every seed draws the number of objects (40..4000), the
cloud's density, frame size, FSAA, recursion depth, Gamma, whether the lists come from the generator or from the
list-building pass, whether tile lists are rebuilt on the GPU, and the grid thresholds; pixels, hit ids and ray counts must
equal the oracle's.  The suite runs QR_SWEEP_COUNT seeds (default 10) from QR_SWEEP_FIRST (default 1000); a longer sweep:
QR_SWEEP_FIRST=5000 QR_SWEEP_COUNT=200 python -m pytest tests/test_random_sweep.py -m gpu -s"""
import os
import random
import time

import numpy as np
import pytest

from test_synth import _synth

FIRST = int(os.environ.get("QR_SWEEP_FIRST", "1000"))
COUNT = int(os.environ.get("QR_SWEEP_COUNT", "10"))


def draw(seed):
    rnd = random.Random(seed)
    n = int(40 * (100 ** rnd.random()))
    box = rnd.choice([6.0, 10.0, 16.0, 30.0, 60.0]) * (n / 500.0) ** (1.0 / 3.0)
    w, h = rnd.choice([(160, 90), (200, 113), (320, 180), (97, 61)])
    kw = dict(n_objects=n, width=w, height=h, depth=rnd.choice([0, 1, 3, 6, 10]), box=box, seed=seed, fsaa=rnd.choice([0, 0, 1, 2]),
              gamma=rnd.random() < 0.3)
    built = rnd.random() < 0.7
    rebin = rnd.random() < 0.5
    low = rnd.random() < 0.5 and n >= 400
    return kw, built, rebin, low


@pytest.mark.gpu
def test_gpu_random_scene_sweep_matches_oracle(qr, oracle, capsys):
    import torch
    bad = []
    for seed in range(FIRST, FIRST + COUNT):
        kw, built, rebin, low = draw(seed)
        t0 = time.time()
        raw = _synth().make_scene(shadow_lists=not built, **kw)
        blob = qr.build_lists(raw) if built else raw
        env = {"QR_DDA": "64", "QR_GRID": "64"} if low else {}
        os.environ.update(env)
        try:
            scn = qr.Scene(blob, rebin_tiles=rebin)
        finally:
            for k in env:
                del os.environ[k]
        frame = scn.new_frame(); ids = torch.full_like(frame, -2)
        scn.render(frame, ids=ids); torch.cuda.synchronize()
        o_frame, o_ids, _ = oracle.render(blob, threads=16, want_ids=True)
        _, _, o_counts = oracle.render(blob, threads=16, deferred=True)
        _, c = scn.render_count()
        dpx = int((frame.cpu().numpy().view(np.uint32) != o_frame).sum())
        did = int((ids.cpu().numpy() != o_ids).sum())
        cnt_ok = c.as_dict() == {k: o_counts[k] for k in c.as_dict()}
        ok = dpx == 0 and did == 0 and cnt_ok
        line = ("seed %d n %d box %.1f %dx%d fsaa %d depth %d gamma %d built %d rebin %d low-thresholds %d: %s (pixels %d ids %d counts %s) %.1fs" % (
            seed, kw["n_objects"], kw["box"], kw["width"], kw["height"], kw["fsaa"], kw["depth"], kw["gamma"], built, rebin, low,
            "equal" if ok else "DIFFERENT", dpx, did, cnt_ok, time.time() - t0))
        if not ok:
            bad.append(seed)
        if not ok or COUNT > 10:
            with capsys.disabled():         # a long sweep shows its progress
                print(line, flush=True)
    with capsys.disabled():
        print("\nrandom sweep: %d scenes from seed %d, %d different" % (COUNT, FIRST, len(bad)), flush=True)
    assert not bad, bad
