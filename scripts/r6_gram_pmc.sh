#!/bin/bash
# PMC passes of the Gram launch (scripts/gram_timing.py 100000: split pre-pass + syrk_tn_split_w4_kernel, or the kernel SDM_GRAM_KERNEL names), one counter set per pass
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/gram_pmc_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python $REPO/scripts/gram_timing.py 100000 > /dev/null 2> $OUT/p${i}_stderr.log
done
python $REPO/scripts/pmc_by_grid.py $OUT syrk_tn_split > $OUT/summary.txt
python $REPO/scripts/pmc_by_grid.py $OUT split_planes >> $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/p*/
