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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 $R/bench.py --steps 100 --warmup 10 --inflight 1 --no-cpu-baseline > $O/${TAG}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/${TAG}_pmc_SQ1 -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmc_SQ1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $O/${TAG}_pmc_SQ2 -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_pmc_SQ2.log 2>&1
# the other BASELINE configurations (parity-test cases; not the bench line): one JSON line each
for w in demo1_1080p_d0 demo2_1080p_gf_d3 demo2_2160p_aa4 synth10k_4320p; do
  python3 $R/bench.py --workload $w --steps 200 --warmup 20 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || true
done
echo done
