#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 200 python tools/gpu_synth_probe.py > $O/r4f_probe.txt 2>&1; cat $O/r4f_probe.txt | grep -v amdgpu
timeout -k 10 300 python tools/gpu_stats.py synth:10000:7680:4320:4 > $O/r4f_stats.txt 2>&1; grep -v amdgpu $O/r4f_stats.txt
