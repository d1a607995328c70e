#!/usr/bin/env python3
"""Keep the sources of whatever build a GPU-box run measures: tools/archive_src.py [OUTDIR]  (default gpurun_out/).

Writes OUTDIR/src_<fingerprint>.tgz -- csrc/*.{hip,hpp,h,cpp}, include/*.h and the Makefile, the files qr_version()'s source
fingerprint is taken over -- unless that archive exists.  The gpu_*.sh tools call it first, so that a measurement (or a fault) of
an intermediate state that never becomes a commit can still be explained from its source (the round-2 memory fault could not)."""
import glob, hashlib, os, sys, tarfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
C = os.path.join(ROOT, "quadray-engine_amd", "csrc")
files = sorted(glob.glob(os.path.join(C, "*.hip")) + glob.glob(os.path.join(C, "*.hpp")) + glob.glob(os.path.join(C, "*.h")) +
               glob.glob(os.path.join(C, "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))) + [os.path.join(C, "Makefile")]
h = hashlib.sha1()
for f in files:                     # the Makefile's SRC_HASH: sha1 of the concatenation in this order
    h.update(open(f, "rb").read())
fp = h.hexdigest()[:12]
os.makedirs(out, exist_ok=True)
path = os.path.join(out, f"src_{fp}.tgz")
if not os.path.exists(path):
    with tarfile.open(path, "w:gz") as t:
        for f in files:
            t.add(f, arcname=os.path.relpath(f, ROOT))
print(f"sources {fp} -> {os.path.relpath(path, ROOT)}")
