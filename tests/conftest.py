import gzip
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


MANIFEST = _manifest()
SMALL_CASES = sorted(n for n, e in MANIFEST.items() if "frame" in e)
BIG_CASES = sorted(n for n, e in MANIFEST.items() if "frame" not in e)


def load_blob(name):
    with open(os.path.join(GOLDEN, MANIFEST[name]["snapshot"]), "rb") as f:
        return gzip.decompress(f.read())


def load_frame(name):
    with open(os.path.join(GOLDEN, MANIFEST[name]["frame"]), "rb") as f:
        return np.load(io.BytesIO(gzip.decompress(f.read())))


@pytest.fixture(scope="session")
def oracle():
    import qr_oracle
    if not os.path.exists(qr_oracle.LIB_PATH):
        import __graft_entry__ as g
        g.build_oracle()
    return qr_oracle


@pytest.fixture(scope="session")
def qr():
    from qr_loader import load_package
    return load_package()
