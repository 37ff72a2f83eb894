#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for F in "27201 136" "8801 44"; do
  set -- $F
  ( timeout 900 python scripts/r5_solve_ab.py $1 $2 4096 - SDM_UPDATE_TPW=2 SDM_UPDATE_TPW=4 SDM_UPDATE_TPW=8 ) > gpurun_out/r5_tpw_ab_$1.log 2>&1
done
tail -n 5 gpurun_out/r5_tpw_ab_*.log
