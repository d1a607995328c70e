#!/usr/bin/env python3
"""Algorithmic work per frame of the bench workloads: rays and fp32 operations.

Counted by the CPU oracle (oracle/qr_oracle.c, FL() weights) on the committed snapshots, in both
shading modes: "eager" = the reference's semantics (every depth-test winner is shaded), "deferred"
= only the final hit of a list walk is shaded (what the HIP backend executes; same pixels).
Deterministic per (snapshot, depth); written to tests/golden/work.json.  Entries named "kernel" (and the whole entry of a
synthetic workload) are the kernel's own count of the same steps, put there by tools/gpu_work.py on a GPU box: kept as they are.

    python tests/golden/make_work.py
"""
import gzip, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import qr_oracle

SNAPS = ["c2b_demo01_1080p", "c2_demo01_1080p_d0", "c3_demo02_1080p_gf_d3", "c4_demo02_2160p_aa4_gf", "swarm_demo01_240_1080p"]
try:
    out = json.load(open(os.path.join(HERE, "work.json")))
except Exception:
    out = {}
for name in SNAPS:
    blob = gzip.decompress(open(os.path.join(HERE, name + ".qrs.gz"), "rb").read())
    rec = out.get(name, {})
    for mode, d in (("eager", False), ("deferred", True)):
        _, _, c = qr_oracle.render(blob, threads=8, deferred=d)
        rays = c["primary"] + c["shadow"] + c["reflect"] + c["refract"]
        rec[mode] = dict(rays=rays, flops=c["flops"], **{k: c[k] for k in ("primary", "shadow", "reflect", "refract")})
        print(name, mode, rec[mode], flush=True)
    out[name] = rec
json.dump(out, open(os.path.join(HERE, "work.json"), "w"), indent=1, sort_keys=True)
