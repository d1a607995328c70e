#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 200 python tools/gpu_synth_probe.py > $O/r4e_probe.txt 2>&1; cat $O/r4e_probe.txt | grep -v amdgpu
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "synth or sweep or random" > $O/r4e_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/r4e_tests.log
