#!/usr/bin/env python3
"""Algorithmic work of bench workloads counted by the kernel itself (QR_PROF build: every lane adds the SURVEY.md 8(d)
weight of each step it executes, the weights oracle/qr_oracle.c applies with FL()).
  tools/gpu_work.py WORKLOAD...   ->  gpurun_out/work_kernel.json     (GPU box; each workload runs in a child process)
Build first: make -C quadray-engine_amd/csrc variant NAME=prof EXTRA=-DQR_PROF"""
import ast, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # imports no GPU code at module level

out = {}
for wl in sys.argv[1:]:
    snap = bench.WORKLOADS[wl][0]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_prof.py"), snap], capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        print(p.stdout[-2000:], p.stderr[-2000:]); raise SystemExit(p.returncode)
    counts = ast.literal_eval([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    flops = int(re.search(r"algorithmic fp32 operations executed .* (\d+)$", p.stderr, re.M).group(1))
    cull = int(re.search(r"operations of the implementation's own culls .* (\d+)$", p.stderr, re.M).group(1))
    prof = {m.group(1).strip(): int(m.group(2)) for m in re.finditer(r"^QR_PROF (.+?)\s+(\d+)$", p.stderr, re.M)}
    rays = sum(counts[k] for k in ("primary", "shadow", "reflect", "refract"))
    out[snap] = dict(workload=wl, flops=flops, rays=rays, primary=counts["primary"], shadow=counts["shadow"],
                     reflect=counts["reflect"], refract=counts["refract"], flops_per_ray=flops / max(1, rays), cull_flops=cull,
                     source="kernel-side count (QR_PROF build of the COUNT kernel instance, one atomic add per step)", prof=prof)
    print(wl, snap, "flops", flops, "rays", rays, "per ray %.1f" % (flops / max(1, rays)), "cull flops", cull, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "work_kernel.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)

# tests/golden/work.json: every entry gets the kernel-side figures beside the oracle's ("kernel": bench.py's flops_kernel); the
# oracle's own figure stays what the oracle counted ("deferred", or "oracle_band" for a scene it cannot walk whole)
wj = os.path.join(ROOT, "tests", "golden", "work.json")
with open(wj) as f:
    work = json.load(f)
for snap, e in out.items():
    k = {x: e[x] for x in ("flops", "rays", "primary", "shadow", "reflect", "refract", "cull_flops", "source")}
    ent = work.setdefault(snap, {})
    ent["kernel"] = k
with open(wj, "w") as f:
    json.dump(work, f, indent=1, sort_keys=True)
