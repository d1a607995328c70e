#!/bin/bash
# usage (GPU box): tools/gpu_dropin_ab.sh LIB...   -- per-call time of the drop-in path (unmodified engine -> qr_render0) on demo scene 1 at
# 1080p, frozen and animated, registered frame, with several builds of libqrhip (LD_PRELOAD), alternating, same box
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
R=${GRAFT_REPO_ROOT:-$PWD}
T=$(mktemp -d); mkdir -p $T/dump; cd $T
for rep in 1 2; do
  for L in "$@"; do
    for mode in "" "--animate 33"; do
      echo -n "$L [${mode:-frozen}] "
      LD_PRELOAD=$R/quadray-engine_amd/$L $R/oracle/_ref/qr_ref_shim --scene demo01 -w 1920 -h 1080 --gpu --bench 50 --pin-frame $mode 2>/dev/null | grep gpu_bench | sed -e 's/gpu_bench frames 50 //' -e 's/ (engine.*//'
    done
  done
done
