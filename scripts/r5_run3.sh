#!/bin/bash
# round 5, GPU call 3: both generations of the chain kernels (stamps, check against float64, A/B inside the solve), the RCR-68 training probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps2.log 2>&1
for v1 in 1 0; do
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_v1_$v1.log 2>&1
  ( SDM_SOLVE_CHAIN_V1=$v1 timeout 600 python scripts/r5_rcr68_train_probe.py 100000 68 ) > gpurun_out/r5_probe68_v1_$v1.log 2>&1
done
( SDM_SOLVE_CHAIN_V1=1 PROBE_TORCH_STREAM=0 timeout 600 python scripts/r5_rcr68_train_probe.py 100000 68 ) > gpurun_out/r5_probe68_own_stream.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_run3_tests.log 2>&1
cat gpurun_out/r5_chain_stamps2.log; tail -n 2 gpurun_out/r5_solve_ab_*_v1_*.log gpurun_out/r5_probe68_*.log; tail -6 gpurun_out/r5_run3_tests.log
