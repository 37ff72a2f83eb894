#!/bin/bash
# round 5, GPU call 2: QR / distributed / exchange tests after the housekeeping; chain stamps; head split with the mid queue at normal / high
# priority; float16 trailing updates from fewer tiles on; two-halves detect A/B; bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 scripts/ubench/bin/chain_stamps 70; timeout 300 scripts/ubench/bin/chain_stamps 213 ) > gpurun_out/r5_chain_stamps.log 2>&1
( timeout 600 python scripts/r5_halves_ab.py 4096 50 ) > gpurun_out/r5_halves_ab.log 2>&1
for prio in n h; do
  ( SDM_SOLVE_MID_PRIO=$prio timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 4,1 ) > gpurun_out/r5_solve_ab_8801_mid_$prio.log 2>&1
done
for mt in 8 16 24; do
  ( SDM_SOLVE_UPD_MIN_TILES=$mt timeout 600 python scripts/r5_solve_ab.py 8801 44 4096 4,0 ) > gpurun_out/r5_solve_ab_8801_mt$mt.log 2>&1
  ( SDM_SOLVE_UPD_MIN_TILES=$mt timeout 600 python scripts/r5_solve_ab.py 27201 136 4096 4,0 ) > gpurun_out/r5_solve_ab_27201_mt$mt.log 2>&1
done
( timeout 1800 python -m pytest tests/test_gpu_qr_solver.py tests/test_gpu_exchange.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r5_run2_tests.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err
cat gpurun_out/r5_chain_stamps.log gpurun_out/r5_halves_ab.log; tail -n 3 gpurun_out/r5_solve_ab_*_mid_*.log gpurun_out/r5_solve_ab_*_mt*.log; tail -8 gpurun_out/r5_run2_tests.log; tail -c 600 gpurun_out/r5_bench1.json; tail -3 gpurun_out/r5_bench1.err
