#!/bin/bash
# Drop-in fuzz on the GPU box: the unmodified engine renders random scenes with its own CPU backend and through
# qr_render0 (oracle/_ref/qr_ref_shim --gpu compares the two frames): transforms fuzzed inside the engine (--jitter)
# and swarms of extra quadrics (--swarm).  usage: tools/gpu_dropin_fuzz.sh [seeds]  -> one line per scene + a summary
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-10}
T=$(mktemp -d); mkdir -p $T/dump; cd $T
ok=0; bad=0; err=0
run() {
  out=$($R/oracle/_ref/qr_ref_shim "$@" --gpu 2>&1)
  if echo "$out" | grep -q "MISMATCH"; then bad=$((bad+1)); echo "MISMATCH: $*"
  elif echo "$out" | grep -q "MATCH"; then ok=$((ok+1))
  else err=$((err+1)); echo "ERROR: $* :: $(echo "$out" | tail -1)"; fi
}
for s in $(seq 1 $N); do
  for sc in demo01 demo02 demo03; do
    run --scene $sc -w 320 -h 240 --swarm 200,$s,1
    run --scene $sc -w 320 -h 240 --swarm 160,$s --gamma --fresnel --fsaa 2 -t $((s*777))
  done
  for sc in test03 test07 test09 test11 test13 test14 test16 demo02; do
    run --scene $sc -w 200 -h 150 --jitter $s
  done
done
echo "drop-in fuzz: $ok scenes MATCH, $bad MISMATCH, $err errors"
