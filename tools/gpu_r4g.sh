#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
QR_STEPS=600 bash tools/gpu_ab.sh libqrhip_noloop.so libqrhip.so demo1_1080p demo2_1080p_gf_d3 2>&1 | tee $O/r4g_ab.txt
QR_STEPS=40 bash tools/gpu_ab.sh libqrhip_noloop.so libqrhip.so synth10k_4320p 2>&1 | tee -a $O/r4g_ab.txt
