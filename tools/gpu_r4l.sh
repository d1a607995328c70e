#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
for rep in 1 2 3; do for L in libqrhip.so libqrhip_kp8.so libqrhip_kp16.so libqrhip_kp8h2.so; do
  for w in demo1_1080p demo2_1080p_gf_d3; do
    QR_LIB=$R/quadray-engine_amd/$L python bench.py --workload $w --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$L', round(d['value'],1), 'Mrays/s', 'isolated', round(d['roofline']['kernel_avg_ms'],4), 'min', round(d['roofline']['kernel_min_ms'],4), d['config']['frame_check']['ok'])" | tee -a $O/r4l_kp.txt
  done; done; done
