#!/bin/bash
# round 5, GPU call 11: the whole -m gpu suite, the round's profiles (kernel stats + PMC passes of the bench command), the per-rank chain of the sharded factorisation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r5_run11_tests.log 2>&1
( timeout 600 python scripts/sharded_solve_timing.py ) > gpurun_out/r5_sharded_timing.log 2>&1
cp gpurun_out/sharded_solve_timing.json gpurun_out/r05_sharded_solve_timing.json 2>/dev/null
( timeout 1500 bash scripts/profile_bench.sh r05 ) > gpurun_out/r5_profile.log 2>&1
tail -8 gpurun_out/r5_run11_tests.log; tail -5 gpurun_out/r5_sharded_timing.log; tail -60 gpurun_out/r5_profile.log
