#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for F in "8801 44" "27201 136" "27201 44"; do
  set -- $F
  ( timeout 600 python scripts/r5_solve_ab.py $1 $2 4096 4,0 ) > gpurun_out/r5_solve_ab_$1_$2_bsx.log 2>&1
done
( timeout 1200 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r5_run14_tests.log 2>&1
( timeout 600 python scripts/r5_rcr68_train_probe.py 100000 68 ) > gpurun_out/r5_probe68_bsx.log 2>&1
tail -n 1 gpurun_out/r5_solve_ab_*_bsx.log gpurun_out/r5_probe68_bsx.log; tail -3 gpurun_out/r5_run14_tests.log
