#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_qr_solver.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r5_qr_tests.log 2>&1
( timeout 900 python scripts/r4_qr_large.py ) > gpurun_out/r5_qr_large.log 2>&1
tail -15 gpurun_out/r5_qr_tests.log; tail -6 gpurun_out/r5_qr_large.log
