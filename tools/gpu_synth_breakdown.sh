#!/bin/bash
# config 5: where the time goes today (stats build, depth probe, knobs)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 300 python tools/gpu_stats.py synth:10000:7680:4320:4 > $O/breakdown_stats.txt 2>&1 && \
timeout -k 10 200 python tools/gpu_synth_probe.py > $O/breakdown_probe.txt 2>&1 && \
for d in 0 1 2; do QR_LIB=$R/quadray-engine_amd/libqrhip_knobs.so QR_DBG=$d timeout -k 10 200 python tools/gpu_synth_probe.py >> $O/breakdown_knobs.txt 2>&1 || break; echo "^ QR_DBG=$d" >> $O/breakdown_knobs.txt; done
cat $O/breakdown_stats.txt $O/breakdown_probe.txt $O/breakdown_knobs.txt | grep -v amdgpu.ids
