#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run4; mkdir -p $O
export SDM_HOG_MODES=2
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
for v in packed plain; do
  if [ $v = plain ]; then export SDM_HOG_NO_PACK=1; fi
  timeout 300 scripts/pmc_cmd.sh r2_${v}_1 "$P1" python $PWD/scripts/gpu_hogtime.py > /dev/null 2>&1
  timeout 300 scripts/pmc_cmd.sh r2_${v}_2 "$P2" python $PWD/scripts/gpu_hogtime.py > /dev/null 2>&1
  python scripts/pmc_by_grid.py gpurun_out/pmc_r2_${v}_1 hog > $O/pmc_${v}.txt
  python scripts/pmc_by_grid.py gpurun_out/pmc_r2_${v}_2 hog >> $O/pmc_${v}.txt
done
cat $O/pmc_packed.txt
