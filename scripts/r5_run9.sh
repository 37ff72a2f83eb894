#!/bin/bash
# round 5, GPU call 9: chain generation 3 final form; full solver / config tests; bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps8.log 2>&1
for v1 in 1 0; do
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 17051 44 4096 4,0 ) > gpurun_out/r5_solve_ab_17051_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_v1_$v1.log 2>&1
done
( timeout 1500 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_exchange.py tests/test_gpu_configs.py tests/test_gpu_qr_solver.py tests/test_gpu_full_size_properties.py -m gpu -q -s 2>&1 | grep -E "distance from|teacher-forced|config 5|passed|failed|FAILED|Error" | tail -30 ) > gpurun_out/r5_run9_tests.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu ) > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err
cat gpurun_out/r5_chain_stamps8.log | sed -n '/generation 2/,$p'; tail -n 1 gpurun_out/r5_solve_ab_*_v1_*.log; cat gpurun_out/r5_run9_tests.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench3.json').read().strip().splitlines()[-1])
print('bench3', d['value'], d['train']['sec_per_cascade'], d['train']['stage_ms_per_level_rank0'], d['rcr68_train']['sec_per_cascade'], d['rcr68_train']['stage_ms_per_level_rank0'])
PY
