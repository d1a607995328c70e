#!/bin/bash
# usage: tools/gpu_bench_quick.sh [workloads...]   (runs on the GPU box)
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
for w in "${@:-demo1_1080p}"; do
  python bench.py --workload $w --steps ${QR_STEPS:-300} --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', 'QR_WAVES=${QR_WAVES:-def}', round(d['value'],1), 'Mrays/s', round(d['roofline']['kernel_avg_ms'],4), 'ms')"
done
