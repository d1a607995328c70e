#!/usr/bin/env python3
"""Run the counting kernel variant once on a golden snapshot (prints QR_STATS lines of stats builds)."""
import sys, os, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qr_loader import load_package
qr = load_package()
blob = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", sys.argv[1] + ".qrs.gz"), "rb").read())
scn = qr.Scene(blob)
if len(sys.argv) > 2: scn.set_depth(int(sys.argv[2]))
_, c = scn.render_count()
print(sys.argv[1], c.as_dict())
