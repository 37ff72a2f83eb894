#!/bin/bash
# PMC passes over the fused detect step (scripts/r4_check_split.py ... t, SDM_R4_ONLY=fused)
export SDM_R4_ONLY=fused
bash scripts/pmc_cmd.sh r4a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" python $PWD/scripts/r4_check_split.py 4096 t
bash scripts/pmc_cmd.sh r4b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" python $PWD/scripts/r4_check_split.py 4096 t
bash scripts/pmc_cmd.sh r4c "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS" python $PWD/scripts/r4_check_split.py 4096 t
