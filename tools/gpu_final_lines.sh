#!/bin/bash
# usage (GPU box): tools/gpu_final_lines.sh TAG -- the bench line of every workload with the committed counters of THIS build beside it
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
python3 bench.py > $O/${TAG}_final_bench_n1.json 2> $O/${TAG}_final_bench_n1.err; tail -c 200 $O/${TAG}_final_bench_n1.json; echo
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_final_bench_steps20.json 2>/dev/null
for w in demo1_1080p_d0 demo2_1080p_gf_d3 demo2_2160p_aa4 swarm_1080p synth10k_4320p; do
  python3 bench.py --workload $w --steps 100 --warmup 10 > $O/${TAG}_final_bench_$w.json 2> $O/${TAG}_final_bench_$w.err || echo "FAILED $w"
done
python3 - <<PY
import json,os
O="$O"
for f in sorted(os.listdir(O)):
    if f.startswith("${TAG}_final_bench_") and f.endswith(".json"):
        try:
            d=json.loads(open(os.path.join(O,f)).read().strip().splitlines()[-1]); r=d["roofline"]
            print(f, round(d["value"],1), "ms/step", round(d["ms_per_step"],5), "iso", round(r["kernel_avg_ms"],4), "basis", r.get("basis"), "frac", round(r["frac"],4) if r.get("frac") else None, "traffic", r.get("traffic"), "lane-ops/flop", r.get("valu_lane_ops_per_flop"), "cpu", round(d["cpu_baseline"]["value"],1) if d.get("cpu_baseline") else None)
        except Exception as e: print(f, "ERR", e)
PY
