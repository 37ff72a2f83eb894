#!/bin/bash
# round 5, GPU call 8: panel-solve loads before LDS writes, refinement in the in-tile panel solve, back-substitution chunking, config-5 fixture
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps7.log 2>&1
( timeout 300 scripts/ubench/bin/chain_stamps_norefine 70 ) > gpurun_out/r5_chain_stamps7_norefine.log 2>&1
for cap in 0 1 3; do
  ( SDM_SOLVE_BS_CAP=$cap timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_cap$cap.log 2>&1
done
( timeout 600 python scripts/r5_solve_ab.py 17051 44 4096 4,0 ) > gpurun_out/r5_solve_ab_17051_g3.log 2>&1
( SDM_SOLVE_BS_CAP=5 timeout 600 python scripts/r5_solve_ab.py 17051 44 4096 4,0 ) > gpurun_out/r5_solve_ab_17051_cap5.log 2>&1
( timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_g3.log 2>&1
( timeout 600 python -m pytest "tests/test_gpu_configs.py::test_teacher_forced_training_level_by_level" -m gpu -q -s 2>&1 | grep -E "distance from|teacher-forced|passed|failed" ) > gpurun_out/r5_tf_refine.log 2>&1
( timeout 1500 python scripts/make_config5_fixture.py 20000 gpurun_out/config5_20k_level0.npz ) > gpurun_out/r5_config5_fixture.log 2>&1
sed -n '/generation 3/,$p' gpurun_out/r5_chain_stamps7.log; sed -n '/generation 3/,$p' gpurun_out/r5_chain_stamps7_norefine.log | head -4; tail -n 1 gpurun_out/r5_solve_ab_*_cap*.log gpurun_out/r5_solve_ab_*_g3.log; cat gpurun_out/r5_tf_refine.log; tail -3 gpurun_out/r5_config5_fixture.log
