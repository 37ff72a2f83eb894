#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python scripts/r5_bs_determinism.py 17051 88 ) > gpurun_out/r5_bs_det_17051.log 2>&1
( timeout 600 python scripts/r5_bs_determinism.py 8801 44 ) > gpurun_out/r5_bs_det_8801.log 2>&1
( timeout 600 python scripts/r5_bs_determinism.py 27201 136 ) > gpurun_out/r5_bs_det_27201.log 2>&1
for F in "8801 44" "17051 88" "27201 136"; do
  set -- $F
  ( timeout 900 python scripts/r5_solve_ab.py $1 $2 4096 - SDM_SOLVE_BS_CAP=5 ) > gpurun_out/r5_bs7_ab_$1.log 2>&1
done
( timeout 1500 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_configs.py tests/test_gpu_qr_solver.py tests/test_gpu_distributed.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r5_bs7_tests.log 2>&1
grep -c identical gpurun_out/r5_bs_det_*.log; grep DIFFERS gpurun_out/r5_bs_det_*.log | head -20
tail -n 3 gpurun_out/r5_bs7_ab_*.log; tail -8 gpurun_out/r5_bs7_tests.log
