#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
T=$(mktemp -d); mkdir -p $T/dump; cd $T
for rep in 1 2 3; do for o in 0 1; do
  for extra in "" "--animate 33"; do
    QR_DROPIN_ORDER=$o $R/oracle/_ref/qr_ref_shim --scene demo01 -w 1920 -h 1080 --gpu --bench 200 $extra 2>/dev/null | grep -E "gpu_bench|MISMATCH" | sed "s/^/order=$o [$extra] /" | tee -a $O/r4o_order.txt
  done
  QR_DROPIN_ORDER=$o $R/oracle/_ref/qr_ref_shim --scene demo02 -w 1920 -h 1080 --gamma --fresnel --gpu --bench 100 2>/dev/null | grep -E "gpu_bench|MISMATCH" | sed "s/^/order=$o [demo02 gf] /" | tee -a $O/r4o_order.txt
done; done
cd $R; timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "drop_in or devices" 2>&1 | tail -3
