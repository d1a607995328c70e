"""CPU suite: the multi-GPU decomposition (tile-row blocks + one exchange per step) on gloo."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,case", [(2, "demo01_160"), (3, "demo02_odd_33x17_aa4"), (2, "demo01_odd_157x93")])
def test_tile_row_sharding_assembles_frames(oracle, world, case):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), case], env=env))
    rcs = [p.wait(timeout=300) for p in procs]
    assert rcs == [0] * world


def test_block_rows_partition():
    import importlib.util
    spec = importlib.util.spec_from_file_location("qr_sharding", os.path.join(ROOT, "quadray-engine_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for h in (1, 7, 8, 9, 93, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            lo = sh.block_rows(h, world)
            assert lo[0] == 0 and lo[-1] == h and all(a <= b for a, b in zip(lo, lo[1:]))
            assert all(x % 8 == 0 for x in lo[:-1])
            for f in range(world):
                assert sorted(sh.block_of(r, f, world) for r in range(world)) == list(range(world))
