#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDM_HOG_MODES=2
timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
for v in abl1 abl2 abl3 abl4 abl5 nomulti; do SDM_HIP_LIB=$PWD/exp/libsdm_$v.so timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"; done
timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
