#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run11; mkdir -p $O
bash scripts/profile_bench.sh r03 > $O/profile_stdout.txt 2>&1
tail -n 60 gpurun_out/profile_r03/summary.txt
cat gpurun_out/profile_r03/summary_full_bench.txt
rm -rf gpurun_out/profile_r03/trace gpurun_out/profile_r03/trace68 gpurun_out/profile_r03/p1 gpurun_out/profile_r03/p2 gpurun_out/profile_r03/p3 gpurun_out/profile_r03/p4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
