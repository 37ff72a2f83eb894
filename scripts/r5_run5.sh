#!/bin/bash
# round 5, GPU call 5: chain kernels generation 3 (accumulator-layout operands, 4-blocked diagonal factor), passes per wave of the pixel kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps4.log 2>&1
for v1 in 1 0; do
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_v1_$v1.log 2>&1
done
( timeout 900 python scripts/r5_detect_env_ab.py SDM_HOG_KPASS 1,2,3,4 4096 50 ) > gpurun_out/r5_kpass_ab.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_exchange.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_run5_tests.log 2>&1
cat gpurun_out/r5_chain_stamps4.log; tail -n 2 gpurun_out/r5_solve_ab_*_v1_*.log; cat gpurun_out/r5_kpass_ab.log; tail -8 gpurun_out/r5_run5_tests.log
