#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_solver_accuracy.py tests/test_gpu_sharded_solve.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r5_newtest.log 2>&1
tail -8 gpurun_out/r5_newtest.log
