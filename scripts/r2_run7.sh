#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run7; mkdir -p $O
export SDM_HOG_MODES=2
timeout 600 python -m pytest tests/test_gpu_packing.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1
tail -n 12 $O/pytest.txt
for rep in 1 2; do
timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
SDM_HIP_LIB=$PWD/exp/libsdm_nomulti.so timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
done
