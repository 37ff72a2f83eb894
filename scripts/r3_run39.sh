#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SDM_GRAM_BF16X3=1 SDM_GRAM_F16X2=1
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s 2>&1 | grep -E "teacher-forced|passed|failed|FAILED|^E  " | head -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_sharded_solve.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | head -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
