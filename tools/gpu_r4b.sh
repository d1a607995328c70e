#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
T=$(mktemp -d); mkdir -p $T/dump; cd $T
for i in 1 2 3 4 5 6 7 8; do
  a=$($R/oracle/_ref/qr_ref_shim --scene test18 -w 128 -h 96 --pt 2 --pt-warm --threads 4 --fsaa 2 | grep "^hash")
  b=$($R/oracle/_ref/qr_ref_shim --scene test18 -w 128 -h 96 --pt 2 --pt-warm --threads 4 --fsaa 2 --shim | grep "^hash")
  c=$(QR_VERIFY=0 $R/oracle/_ref/qr_ref_shim --scene test18 -w 128 -h 96 --pt 2 --pt-warm --threads 4 --fsaa 2 --shim | grep "^hash")
  echo "$i cpu $a | gpu $b | gpu noverify $c" | tee -a $O/r4b_pt.log
done
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -k "two_rank or devices or drop_in_path" > $O/r4b_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/r4b_tests.log
