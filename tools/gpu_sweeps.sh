#!/bin/bash
# round 4, final build: long random sweep (product library, then the guarded one), drop-in fuzz, path-tracer drop-in fuzz
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
PART=${1:-sweep}
if [ $PART = sweep ]; then
  QR_SWEEP_FIRST=${2:-20000} QR_SWEEP_COUNT=${3:-900} timeout -k 10 1100 python -m pytest tests/test_random_sweep.py -m gpu -s -q 2>&1 | tee $O/r4_sweep_${2:-20000}.txt | grep -E "DIFFERENT|random sweep|passed|failed|seed [0-9]*[05]0 "
elif [ $PART = guard ]; then
  QR_LIB=$R/quadray-engine_amd/libqrhip_guard.so QR_SWEEP_FIRST=${2:-30000} QR_SWEEP_COUNT=${3:-400} timeout -k 10 1100 python -m pytest tests/test_random_sweep.py -m gpu -s -q 2>&1 | tee $O/r4_sweep_guard_${2:-30000}.txt | grep -E "DIFFERENT|QR_GUARD|random sweep|passed|failed|seed [0-9]*[05]0 "
else
  timeout -k 10 700 bash tools/gpu_dropin_fuzz.sh ${2:-24} 2>&1 | tee $O/r4_dropin_fuzz.txt | tail -4
  timeout -k 10 300 bash tools/gpu_dropin_pt_fuzz.sh 2>&1 | tee $O/r4_dropin_pt_fuzz.txt | tail -3
fi
