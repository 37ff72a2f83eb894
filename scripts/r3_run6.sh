#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run6; mkdir -p $O
timeout 300 python scripts/solve_ab.py 1300 44 512 2>&1 | tee $O/solve_ab.txt
timeout 300 python scripts/solve_ab.py 8801 44 2>&1 | tee -a $O/solve_ab.txt
timeout 300 python scripts/solve_ab.py 17051 44 2>&1 | tee -a $O/solve_ab.txt
timeout 300 python scripts/solve_ab.py 27201 136 2>&1 | tee -a $O/solve_ab.txt
timeout 1200 python -m pytest tests/test_gpu_solver_accuracy.py tests/test_gpu_sharded_solve.py tests/test_gpu_exchange.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tee $O/pytest.txt
