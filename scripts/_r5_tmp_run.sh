#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
P=$PWD/superviseddescent_amd/lib/libsdm_hip_prio.so
for F in "8801 44" "17051 88" "27201 136"; do
  set -- $F
  ( timeout 900 python scripts/r5_solve_ab.py $1 $2 4096 - SDM_HIP_LIB=$P - SDM_HIP_LIB=$P ) > gpurun_out/r5_prio_ab_$1.log 2>&1
done
tail -n 5 gpurun_out/r5_prio_ab_*.log | cut -c1-200
