#!/usr/bin/env python3
"""Assemble profiles/ from one tools/gpu_profile_round.sh run: tools/make_profiles.py TAG [ROUND]
(reads gpurun_out/TAG_*, writes profiles/<ROUND>_final_demo1_1080p.txt, <ROUND>_bench_n1.json, traffic.json, valu.json)."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
KERN = "qr_render_kernel<false"

def counters(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(G, d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERN in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

def kernel_stats(d):
    for f in glob.glob(os.path.join(G, d, "**", "*_kernel_stats.csv"), recursive=True):
        return [l.rstrip("\n") for l in open(f)][:3]
    return []

bench = open(os.path.join(G, f"{tag}_bench_n1.json")).read().strip()
open(os.path.join(P, f"{rnd}_bench_n1.json"), "w").write(bench + "\n")
b = json.loads(bench)
c = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ1", "pmc_SQ2"):
    c.update(counters(f"{tag}_{d}"))
out = [f"# {rnd} final profile, demo1_1080p (BASELINE metric workload), default build\n",
       "# tools/gpu_profile_round.sh on one MI355X: bench.py; rocprofv3 --kernel-trace --stats; separate --pmc passes\n",
       "# (FETCH_SIZE, WRITE_SIZE, two SQ passes); per-launch means over the qr_render_kernel<false,4> dispatches\n\n"]
out += [l + "\n" for l in kernel_stats(f"{tag}_trace")]
out.append("\n")
for k in sorted(c):
    out.append(f"{k:28s} {c[k]:18.1f}\n")
out.append(f"\n# bench.py line of the same run (HIP-event kernel time {b['roofline']['kernel_avg_ms']*1e3:.1f} us)\n{bench}\n")
open(os.path.join(P, f"{rnd}_final_demo1_1080p.txt"), "w").write("".join(out))
tr = json.load(open(os.path.join(P, "traffic.json")))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    tr["demo1_1080p"] = {"fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
                         "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}
    json.dump(tr, open(os.path.join(P, "traffic.json"), "w"), indent=1)
if "SQ_WAVE_CYCLES" in c and "SQ_INSTS_VALU" in c:
    va = json.load(open(os.path.join(P, "valu.json")))
    w = c["SQ_WAVE_CYCLES"]
    va["demo1_1080p"] = dict(insts_valu=c["SQ_INSTS_VALU"], insts_salu=c["SQ_INSTS_SALU"], insts_smem=c["SQ_INSTS_SMEM"],
                             waves=c.get("SQ_WAVES"), lane_utilisation=c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_INSTS_VALU"]),
                             wave_active_frac=c["SQ_ACTIVE_INST_ANY"] / w, wave_wait_frac=c["SQ_WAIT_ANY"] / w,
                             wave_issue_stall_frac=c["SQ_WAIT_INST_ANY"] / w,
                             valu_issue_quadcycles_per_simd=c["SQ_ACTIVE_INST_VALU"] / 1024,
                             source=f"profiles/{rnd}_final_demo1_1080p.txt")
    json.dump(va, open(os.path.join(P, "valu.json"), "w"), indent=1, sort_keys=True)
import shutil
for f in glob.glob(os.path.join(G, f"{tag}_bench_*.json")):
    name = os.path.basename(f)[len(tag) + 1:]
    if name != "bench_n1.json" and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, f"{rnd}_{name}"))
print("".join(out[:12]))
