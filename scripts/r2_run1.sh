#!/bin/bash
# round 2, GPU call 1: microbenchmarks (kept evidence), HOG phase profile, experiment variants, PMC for LDS conflicts
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run1; mkdir -p $O
for b in valu_rates hbm_read sqrt_domain; do timeout 120 scripts/ubench/bin/$b > $O/ubench_$b.txt 2>&1; done
timeout 300 python scripts/gpu_hogprof.py > $O/hogprof.txt 2>&1
export SDM_HOG_MODES=2
timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_default.txt 2>&1
for v in base st64 st64bb; do SDM_HIP_LIB=$PWD/exp/libsdm_$v.so timeout 200 python scripts/gpu_hogtime.py > $O/hogtime_$v.txt 2>&1; done
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
SDM_HIP_LIB=$PWD/exp/libsdm_base.so timeout 300 scripts/pmc_cmd.sh r2_base "$PMC" python $PWD/scripts/gpu_hogtime.py > $O/pmc_base.txt 2>&1
SDM_HIP_LIB=$PWD/exp/libsdm_st64bb.so timeout 300 scripts/pmc_cmd.sh r2_st64bb "$PMC" python $PWD/scripts/gpu_hogtime.py > $O/pmc_st64bb.txt 2>&1
tail -3 $O/hogtime_*.txt $O/pytest_gpu.txt
