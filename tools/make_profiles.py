#!/usr/bin/env python3
"""Assemble profiles/ from one tools/gpu_profile_round.sh run: tools/make_profiles.py TAG ROUND
(reads gpurun_out/TAG_*, writes profiles/<ROUND>_*.txt / .json and profiles/counters.json)."""
import csv, glob, json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
KERN = "qr_render_kernel<false"
git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()


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


def overlap(d):
    """From the kernel trace: how many render dispatches are in flight at once (dispatch intervals overlap)."""
    iv = []
    for f in glob.glob(os.path.join(G, d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERN in r["Kernel_Name"]:
                iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    if len(iv) < 10:
        return None
    iv = iv[len(iv) // 10:]                      # skip warm-up
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    cur = 0; t_prev = ev[0][0]; hist = collections.Counter()
    for t, dlt in ev:
        hist[cur] += t - t_prev; t_prev = t; cur += dlt
    tot = sum(v for k, v in hist.items() if k > 0)
    span = iv[-1][1] - iv[0][0]
    dur = sum(e - s for s, e in iv) / len(iv)
    return dict(dispatches=len(iv), mean_dispatch_us=dur / 1e3, span_per_dispatch_us=span / len(iv) / 1e3,
                time_with_n_in_flight={str(k): round(v / tot, 3) for k, v in sorted(hist.items()) if k > 0})


bench = open(os.path.join(G, f"{tag}_bench_n1.json")).read().strip().splitlines()[-1]
open(os.path.join(P, f"{rnd}_bench_n1.json"), "w").write(bench + "\n")
b = json.loads(bench)
lib = b.get("lib")            # qr_version() of the build every pass of this run used: bench.py quotes counters only beside the same build

out = [f"# {rnd} profile of the BASELINE metric workload demo1_1080p, default build, git {git}\n",
       "# tools/gpu_profile_round.sh on one MI355X: rocprofv3 --kernel-trace --stats (serial launches, --inflight 1), separate --pmc passes;\n",
       "# per-launch means over the qr_render_kernel<false,4> dispatches\n\n"]
out += [l + "\n" for l in kernel_stats(f"{tag}_trace_serial")]
c = {}
for d in ("pmc_FETCH_SIZE_demo1_1080p", "pmc_WRITE_SIZE_demo1_1080p", "pmc_SQ_demo1_1080p", "pmc_SQ1_demo1_1080p"):
    c.update(counters(f"{tag}_{d}"))
out.append("\n")
for k in sorted(c):
    out.append(f"{k:28s} {c[k]:18.1f}\n")
out.append(f"\n# bench.py line of the same run (HIP-event kernel time {b['roofline']['kernel_avg_ms']*1e3:.1f} us)\n{bench}\n")
open(os.path.join(P, f"{rnd}_final_demo1_1080p.txt"), "w").write("".join(out))

ov = overlap(f"{tag}_trace_inflight3")
o2 = [f"# {rnd} kernel trace of the DEFAULT bench mode (three steps in flight, one HIP stream each), git {git}\n",
      "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline\n\n"]
o2 += [l + "\n" for l in kernel_stats(f"{tag}_trace_inflight3")]
o2.append("\n# dispatch intervals of qr_render_kernel<false,4> from the trace (Start/End timestamps):\n")
o2.append(json.dumps(ov, indent=1) + "\n")
o2.append("# mean_dispatch_us = duration of one launch while others run beside it; span_per_dispatch_us = wall time per frame;\n"
          "# time_with_n_in_flight = share of the busy time with n render dispatches executing at once\n"
          "# NOTE: this is the launch pipeline UNDER THE PROFILER (rocprofv3 serialises dispatches from different streams: ~280 us per\n"
          f"# dispatch).  The same mode without the profiler is profiles/{rnd}_timeline_events_demo1_1080p.txt (HIP events around every launch)\n"
          f"# and profiles/{rnd}_timeline_waves_demo1_1080p.txt (s_memrealtime stamps of every wave): ~44 us per frame, three launches resident.\n")
open(os.path.join(P, f"{rnd}_kernel_trace_inflight3.txt"), "w").write("".join(o2))

cj = {}
for w in ("demo1_1080p", "demo1_1080p_d0", "demo2_1080p_gf_d3", "demo2_2160p_aa4", "synth10k_4320p"):
    cc = {}
    for d in (f"pmc_FETCH_SIZE_{w}", f"pmc_WRITE_SIZE_{w}", f"pmc_SQ_{w}"):
        cc.update(counters(f"{tag}_{d}"))
    if "SQ_INSTS_VALU" not in cc:
        continue
    e = dict(insts_valu=cc["SQ_INSTS_VALU"], insts_salu=cc["SQ_INSTS_SALU"], insts_smem=cc["SQ_INSTS_SMEM"],
             insts_vmem_rd=cc.get("SQ_INSTS_VMEM_RD"), insts_vmem_wr=cc.get("SQ_INSTS_VMEM_WR"),
             waves=cc.get("SQ_WAVES"), wave_cycles=cc.get("SQ_WAVE_CYCLES"),
             thread_cycles_valu=cc.get("SQ_THREAD_CYCLES_VALU"),
             lane_utilisation=cc["SQ_THREAD_CYCLES_VALU"] / (64 * cc["SQ_INSTS_VALU"]) if cc.get("SQ_THREAD_CYCLES_VALU") else None,
             source=f"profiles/{rnd}_counters_all_workloads.txt (rocprofv3 --pmc passes of tools/gpu_profile_round.sh)", git=git, lib=lib)
    if "FETCH_SIZE" in cc and "WRITE_SIZE" in cc:
        # MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 -> doubled; both in KiB
        e.update(fetch_size_kib=cc["FETCH_SIZE"], write_size_kib=cc["WRITE_SIZE"],
                 hbm_bytes_per_launch=int((2 * cc["FETCH_SIZE"] + cc["WRITE_SIZE"]) * 1024))
    cj[w] = e
json.dump(cj, open(os.path.join(P, "counters.json"), "w"), indent=1, sort_keys=True)
lines = [f"# {rnd} per-launch hardware counters of qr_render_kernel<false,4>, all workloads, git {git} (tools/gpu_profile_round.sh)\n"]
for w, e in cj.items():
    lines.append(w + ": " + ", ".join(f"{k} {v:.4g}" if isinstance(v, float) else f"{k} {v}" for k, v in e.items() if k not in ("source", "git", "lib")) + "\n")
open(os.path.join(P, f"{rnd}_counters_all_workloads.txt"), "w").write("".join(lines))

# kernel-side work counts of this build (tools/gpu_work.py on the GPU box writes gpurun_out/work_kernel.json): into work.json
wk = os.path.join(G, "work_kernel.json")
if os.path.exists(wk):
    wj = os.path.join(ROOT, "tests", "golden", "work.json")
    work = json.load(open(wj)); new = json.load(open(wk))
    for snap, e in new.items():
        work.setdefault(snap, {})["kernel"] = {x: e[x] for x in ("flops", "rays", "primary", "shadow", "reflect", "refract", "cull_flops", "source")}
    json.dump(work, open(wj, "w"), indent=1, sort_keys=True)
    shutil_copy = os.path.join(P, f"{rnd}_work_kernel_side.json")
    json.dump(new, open(shutil_copy, "w"), indent=1, sort_keys=True)
for extra in ("timeline_events_demo1_1080p.txt", "timeline_waves_demo1_1080p.txt", "bench_steps20_a.json", "bench_steps20_b.json"):
    src = os.path.join(G, f"{tag}_{extra}")
    if os.path.exists(src) and os.path.getsize(src) > 0:
        open(os.path.join(P, f"{rnd}_{extra}"), "w").write(open(src).read())

import shutil
for f in glob.glob(os.path.join(G, f"{tag}_bench_*.json")):
    name = os.path.basename(f)[len(tag) + 1:]
    if name != "bench_n1.json" and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, f"{rnd}_{name}"))
print("".join(out[:14])); print("".join(o2))
