#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run2; mkdir -p $O
export SDM_HOG_MODES=2
timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_default.txt 2>&1
SDM_HIP_LIB=$PWD/exp/libsdm_pad0.so timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_pad0.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
timeout 300 scripts/pmc_cmd.sh r2_pad1 "$PMC" python $PWD/scripts/gpu_hogtime.py > $O/pmc_pad1.txt 2>&1
grep -h "mode 2" $O/hogtime_*.txt; tail -n 3 $O/pytest_gpu.txt; grep -A9 "^hog_fast" $O/pmc_pad1.txt
