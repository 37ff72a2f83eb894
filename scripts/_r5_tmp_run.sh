#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r5_final_tests.log 2>&1
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r5_final_smoke.log 2>&1
tail -c 300 gpurun_out/r5_final_bench.json; tail -5 gpurun_out/r5_final_tests.log; tail -2 gpurun_out/r5_final_smoke.log
