#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run3; mkdir -p $O
export SDM_HOG_MODES=2
timeout 600 python -m pytest tests/test_gpu_packing.py -x -q > $O/pytest_packing.txt 2>&1
tail -n 15 $O/pytest_packing.txt
timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_packed.txt 2>&1
SDM_HOG_NO_PACK=1 timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_plain.txt 2>&1
grep -h "mode 2" $O/hogtime_*.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -n 8 $O/pytest_gpu.txt
