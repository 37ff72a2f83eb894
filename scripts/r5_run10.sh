#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_qr_solver.py -m gpu -q -x 2>&1 | tail -60 ) > gpurun_out/r5_run10_qr.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_bs.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_bs.log 2>&1
cat gpurun_out/r5_run10_qr.log; tail -n 1 gpurun_out/r5_solve_ab_*_bs.log
