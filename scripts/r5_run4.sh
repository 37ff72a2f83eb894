#!/bin/bash
# round 5, GPU call 4: chain kernels generation 2 with the rolled step loop and the four-wave panel solve
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps3.log 2>&1
for v1 in 1 0; do
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_v1_$v1.log 2>&1
done
( timeout 900 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_exchange.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_run4_tests.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu ) > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err
cat gpurun_out/r5_chain_stamps3.log; tail -n 2 gpurun_out/r5_solve_ab_*_v1_*.log; tail -6 gpurun_out/r5_run4_tests.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench2.json').read().strip().splitlines()[-1])
print('bench2', d['value'], d['train']['stage_ms_per_level_rank0'], d['rcr68_train']['stage_ms_per_level_rank0'])
PY
