#!/bin/bash
# usage (GPU box): tools/gpu_ab.sh LIB_A LIB_B [workloads...]  -- alternating runs of two builds of libqrhip (QR_LIB), same box, same call
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
for w in "${@:-demo1_1080p}"; do
  for rep in 1 2 3; do
    for L in $A $B; do
      QR_LIB=$R/quadray-engine_amd/$L python $R/bench.py --workload $w --steps ${QR_STEPS:-300} --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$L', round(d['value'],1), 'Mrays/s', 'isolated', round(d['roofline']['kernel_avg_ms'],4), 'min', round(d['roofline']['kernel_min_ms'],4))"
    done
  done
done
