#!/bin/bash
# round 5, GPU call 7: panel-solve staging fixed; which change moved the float64 distance of RCR-22 level 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps6.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_g3.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_g3.log 2>&1
for cfg in "0 16" "1 16" "0 40" "1 40" "0 24"; do
  set -- $cfg
  ( SDM_SOLVE_CHAIN_V1=$1 SDM_SOLVE_UPD_MIN_TILES=$2 timeout 600 python -m pytest "tests/test_gpu_configs.py::test_teacher_forced_training_level_by_level" -m gpu -q -s 2>&1 | grep -E "distance from|teacher-forced|passed|failed" ) > gpurun_out/r5_tf_v1_$1_mt_$2.log 2>&1
done
sed -n '/generation 3/,$p' gpurun_out/r5_chain_stamps6.log; tail -n 1 gpurun_out/r5_solve_ab_*_g3.log; for f in gpurun_out/r5_tf_v1_*; do echo $f; cat $f; done
