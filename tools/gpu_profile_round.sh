#!/bin/bash
# usage (GPU box): tools/gpu_profile_round.sh TAG  -> gpurun_out/TAG_*  (bench line, kernel trace stats, HBM counters)
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
tail -c 600 $O/${TAG}_bench_n1.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/${TAG}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmc_$c.log 2>&1
done
echo done
