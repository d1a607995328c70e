#!/bin/bash
# usage (GPU box): tools/gpu_band_ab.sh NAME y0 y1 LIB...  -- launch time of one band of rows rendered alone, per build of libqrhip
python3 ${GRAFT_REPO_ROOT:-$PWD}/tools/archive_src.py >/dev/null 2>&1 || true
N=$1; Y0=$2; Y1=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do for L in "$@"; do echo -n "$L: "; QR_LIB=$R/quadray-engine_amd/$L python $R/tools/gpu_band.py $N $Y0 $Y1 40 2>/dev/null | tail -n 1; done; done
