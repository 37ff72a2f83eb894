#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_colour_gray.py -x -q -m gpu > $O/pytest_new.txt 2>&1
tail -n 25 $O/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt
