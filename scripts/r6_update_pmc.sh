#!/bin/bash
# PMC passes of one factor + solve at F = 27 201 (scripts/r5_solve_ab.py --child): matrix-pipe busy cycles, clocks and L2 behaviour of
# the trailing update's launches (syrk_update_f16_w4_kernel), per grid size.  $1 = a label; SDM_SOLVE_UPD_WIDE=1 in the environment:
# the one-wave-per-SIMD stream.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/update_pmc_r6${1:+_$1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python $REPO/scripts/r5_solve_ab.py --child 27201 136 4096 /tmp/x.npy > /dev/null 2> $OUT/p${i}_stderr.log
done
python $REPO/scripts/pmc_by_grid.py $OUT syrk_update_f16_w4 > $OUT/summary.txt
rm -rf $OUT/p*/
