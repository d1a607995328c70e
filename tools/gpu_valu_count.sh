#!/bin/bash
# usage (GPU box): tools/gpu_valu_count.sh [workload...]  -> per-frame SQ instruction counts (deterministic) + kernel time
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for w in "${@:-demo1_1080p}"; do
  rm -rf $R/gpurun_out/valu_$w
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/valu_$w -o p -- python3 $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/valu_$w.log 2>&1
  python3 $R/tools/prof_summary.py $R/gpurun_out/valu_$w "qr_render_kernel<false" | grep -v "^#\|^\"" | awk -v w=$w '{printf "%s %s %.2fM  ", w, $1, $3/1e6} END {print ""}'
done
