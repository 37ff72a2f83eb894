#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for F in "8801 44" "17051 88" "27201 136"; do
  set -- $F
  ( timeout 900 python scripts/r5_solve_ab.py $1 $2 4096 - SDM_SOLVE_SPLIT_TRSM=1 - SDM_SOLVE_SPLIT_TRSM=1 ) > gpurun_out/r5_split_ab_$1.log 2>&1
done
tail -n 5 gpurun_out/r5_split_ab_*.log
