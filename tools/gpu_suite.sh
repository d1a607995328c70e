#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
python3 $R/tools/archive_src.py >/dev/null 2>&1 || true
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite_tests.log 2>&1; echo "tests rc $?"; tail -25 $O/suite_tests.log
