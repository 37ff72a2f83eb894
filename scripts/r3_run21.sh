#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run21; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2; grep -E "^FAILED" $O/pytest_gpu.txt | head -8
timeout 600 python scripts/sharded_solve_timing.py rcr22 rcr68 > $O/sharded.txt 2>&1; tail -n 12 $O/sharded.txt
cp gpurun_out/sharded_solve_timing.json $O/ 2>/dev/null
bash scripts/profile_bench.sh r03 > $O/profile_stdout.txt 2>&1
rm -rf gpurun_out/profile_r03/trace gpurun_out/profile_r03/trace68 gpurun_out/profile_r03/p1 gpurun_out/profile_r03/p2 gpurun_out/profile_r03/p3 gpurun_out/profile_r03/p4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
