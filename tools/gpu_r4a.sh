#!/bin/bash
# round 4, first GPU call: suite, bench (default and the driver's --steps 20), non-profiler timelines, issue-rate micro-benchmark
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/r4a_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/r4a_tests.log
timeout -k 10 200 python bench.py > $O/r4a_bench.json 2> $O/r4a_bench.err && tail -c 600 $O/r4a_bench.json && \
timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4a_bench20.json 2> $O/r4a_bench20.err && \
timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4a_bench20b.json 2>> $O/r4a_bench20.err && \
timeout -k 10 100 python tools/gpu_timeline.py events 96 > $O/r4a_tl_events.log 2>&1 && \
timeout -k 10 150 python tools/gpu_timeline.py waves 96 > $O/r4a_tl_waves.log 2>&1 && \
timeout -k 10 100 tools/ubench/issue_rate > $O/r4a_issue_rate.txt 2>&1
echo "rc $?"
python3 - <<'PY'
import json,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out")
for f in ("r4a_bench.json","r4a_bench20.json","r4a_bench20b.json"):
    try:
        d=json.loads(open(os.path.join(O,f)).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), "ms/step", round(d["ms_per_step"],5), d["timing"]["repetitions"], round(d["timing"]["timed_region_ms"],1), round(d["timing"]["ms_per_step_region_mean"],5), round(d["timing"]["ms_per_step_min_rep"],5), round(d["timing"]["ms_per_step_max_rep"],5), "iso", round(d["roofline"]["kernel_avg_ms"],5))
    except Exception as e: print(f, "ERR", e)
PY
tail -8 $O/r4a_tl_events.log; tail -8 $O/r4a_tl_waves.log; cat $O/r4a_issue_rate.txt
