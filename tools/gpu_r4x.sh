#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "synth or sweep or random" 2>&1 | tail -3
for rep in 1 2 3; do for L in libqrhip_pf0.so libqrhip.so; do
  QR_LIB=$R/quadray-engine_amd/$L python bench.py --workload synth10k_4320p --steps 30 --warmup 5 --no-cpu-baseline --repetitions 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), 'Mrays/s', 'isolated', round(d['roofline']['kernel_avg_ms'],4))" | tee -a $O/r4p_prefetch.txt
done; done
