#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
for rep in 1 2 3; do for x in 0 1; do
  for w in demo1_1080p demo2_1080p_gf_d3 demo2_2160p_aa4 synth10k_4320p; do
    st=300; [ $w = synth10k_4320p ] && st=30; [ $w = demo2_2160p_aa4 ] && st=60
    QR_SCHED_XCD=$x python bench.py --workload $w --steps $st --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', 'xcd=$x', round(d['value'],1), 'Mrays/s', 'isolated', round(d['roofline']['kernel_avg_ms'],4), 'min', round(d['roofline']['kernel_min_ms'],4))" | tee -a $O/r4j_xcd.txt
  done; done; done
