#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDM_HOG_MODES=2
timeout 600 python -m pytest tests/test_gpu_packing.py tests/test_gpu_parity.py -x -q 2>&1 | tail -n 3
for rep in 1 2 3; do
timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
SDM_HIP_LIB=$PWD/exp/libsdm_head.so timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
done
