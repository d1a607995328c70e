#!/bin/bash
# Path-tracer mode through the drop-in boundary, many configurations (GPU box): N accumulated frames by the engine's own
# CPU backend in one process, by qr_render0 (--shim: every frame) in another; the N-th frames must have the same hash.
R=${GRAFT_REPO_ROOT:-$PWD}
T=$(mktemp -d); mkdir -p $T/dump; cd $T
ok=0; bad=0
for cfg in "96 64 1 10 0 1" "160 120 2 10 0 1" "160 120 3 7 2 1" "200 150 2 10 4 1" "128 96 5 10 0 4" "333 211 2 10 0 3" "160 120 4 3 4 2" \
           "64 48 8 10 0 1" "256 144 2 9 2 1" "160 120 2 0 0 1" "100 100 3 5 4 1" "320 240 2 10 0 1" "161 97 2 10 2 2" "48 200 3 10 0 1" \
           "160 120 6 6 0 1" "192 108 2 10 4 4" "80 60 16 10 0 1" "240 135 3 8 2 1" "120 90 2 2 4 1" "400 300 1 10 0 1"; do
  set -- $cfg
  args="--scene test18 -w $1 -h $2 --pt $3 --depth $4 --fsaa $5 --threads $6"
  [ $(( ($1 + $2) % 3 )) = 0 ] && args="$args --gamma --fresnel"
  a=$($R/oracle/_ref/qr_ref_shim $args 2>&1 | grep "^hash" | head -1)
  b=$($R/oracle/_ref/qr_ref_shim $args --shim 2>&1 | grep "^hash" | head -1)
  if [ -n "$a" ] && [ "$a" = "$b" ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "MISMATCH: $args :: $a / $b"; fi
done
echo "drop-in path-tracer fuzz: $ok configurations equal, $bad different"
