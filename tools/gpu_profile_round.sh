#!/bin/bash
# usage (GPU box): tools/gpu_profile_round.sh TAG  -> gpurun_out/TAG_*  (bench lines, kernel traces, PMC passes)
# then, in the build container: python tools/make_profiles.py TAG rNN  (writes profiles/rNN_* and profiles/counters.json)
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== bench (default line)"; python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; tail -c 300 $O/${TAG}_bench_n1.json; echo
echo "== kernel trace, default mode (three steps in flight)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_inflight3 -o t -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --repetitions 1 > $O/${TAG}_trace_inflight3.log 2>&1
echo "== kernel trace, serial launches"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_serial -o t -- python3 $R/bench.py --steps 200 --warmup 20 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/${TAG}_trace_serial.log 2>&1
for w in demo1_1080p demo1_1080p_d0 demo2_1080p_gf_d3 demo2_2160p_aa4 synth10k_4320p; do
  st=6; [ $w = synth10k_4320p ] && st=3
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $w"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_${c}_$w -o p -- python3 $R/bench.py --workload $w --steps $st --warmup 2 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/${TAG}_pmc_${c}_$w.log 2>&1
  done
  echo "== pmc SQ $w"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $O/${TAG}_pmc_SQ_$w -o p -- python3 $R/bench.py --workload $w --steps $st --warmup 2 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/${TAG}_pmc_SQ_$w.log 2>&1
done
echo "== pmc SQ busy demo1"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_pmc_SQ1_demo1_1080p -o p -- python3 $R/bench.py --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --repetitions 1 > $O/${TAG}_pmc_SQ1_demo1_1080p.log 2>&1
# the other BASELINE configurations (parity-test cases; not the bench line): one JSON line each
for w in demo1_1080p_d0 demo2_1080p_gf_d3 demo2_2160p_aa4 synth10k_4320p; do
  echo "== bench $w"
  python3 $R/bench.py --workload $w --steps 100 --warmup 10 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || true
done
echo "== non-profiler timelines of the default mode (tools/gpu_timeline.py)"
cd $R
python3 tools/gpu_timeline.py events 96 > $O/${TAG}_timeline_events.log 2>&1 && cp $O/timeline_events_demo1_1080p.txt $O/${TAG}_timeline_events_demo1_1080p.txt
[ -f quadray-engine_amd/libqrhip_wt.so ] && python3 tools/gpu_timeline.py waves 96 > $O/${TAG}_timeline_waves.log 2>&1 && cp $O/timeline_waves_demo1_1080p.txt $O/${TAG}_timeline_waves_demo1_1080p.txt
echo "== the driver's own command line (--steps 20), twice"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_steps20_a.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_steps20_b.json 2>/dev/null
[ -f quadray-engine_amd/libqrhip_prof.so ] && { echo "== kernel-side work counts (QR_PROF)"; python3 tools/gpu_work.py demo1_1080p demo1_1080p_d0 demo2_1080p_gf_d3 demo2_2160p_aa4 swarm_1080p synth10k_4320p > $O/${TAG}_work.log 2>&1; tail -6 $O/${TAG}_work.log; }
echo done
