#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run37; mkdir -p $O
export SDM_GRAM_BF16X3=1
for np in 3 2; do
  echo "== planes $np"
  SDM_GRAM_PLANES=$np timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -x 2>&1 | grep -E "teacher-forced|passed|failed|FAILED|assert" | head -12
  SDM_GRAM_PLANES=$np timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_sharded_solve.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED" | head -5
  SDM_GRAM_PLANES=$np timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
done
