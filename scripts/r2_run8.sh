#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run8; mkdir -p $O
export SDM_HOG_MODES=2
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -n 4 $O/pytest.txt
for rep in 1 2; do timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"; done
SDM_HOG_NO_PACK=1 timeout 200 python scripts/gpu_hogtime.py 2>&1 | grep "mode 2"
