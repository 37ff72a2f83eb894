#!/bin/bash
# round 5, GPU call 1: solver tests with the head split, schedule A/B, RCR-68 detect A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_run1_tests.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 4,1 2,0 8,0 8,1 ) > gpurun_out/r5_solve_ab_8801.log 2>&1
( timeout 900 python scripts/r5_solve_ab.py 27201 136 4096 4,0 4,1 8,0 8,1 ) > gpurun_out/r5_solve_ab_27201.log 2>&1
( timeout 600 python scripts/r5_rcr68_detect_ab.py 8192 ) > gpurun_out/r5_rcr68_detect_ab.log 2>&1
tail -5 gpurun_out/r5_run1_tests.log; cat gpurun_out/r5_solve_ab_8801.log gpurun_out/r5_solve_ab_27201.log; tail -60 gpurun_out/r5_rcr68_detect_ab.log
