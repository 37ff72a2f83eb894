#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run12; mkdir -p $O
for v in 0 1 0 1; do echo "SDM_GRAM_W8=$v"; SDM_GRAM_W8=$v python scripts/gram_timing.py 100000 2>&1 | tail -n 2; done | tee $O/gram_w8.txt
SDM_GRAM_W8=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "gram or Gram or spd or teacher or train" 2>&1 | tail -n 3 | tee -a $O/gram_w8.txt
