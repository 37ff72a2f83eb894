#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q > $O/pytest_configs.txt 2>&1
tail -n 12 $O/pytest_configs.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 6000 $O/bench.json; tail -n 5 $O/bench.err
