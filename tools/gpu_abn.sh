#!/bin/bash
# usage (GPU box): tools/gpu_abn.sh WORKLOAD LIB...  -- alternating bench runs of several builds of libqrhip (QR_LIB), same box, same call
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
W=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for L in "$@"; do
    QR_LIB=$R/quadray-engine_amd/$L python $R/bench.py --workload $W --steps ${QR_STEPS:-30} --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W', '$L', round(d['value'],1), 'Mrays/s', 'isolated', round(d['roofline']['kernel_avg_ms'],4))"
  done
done
