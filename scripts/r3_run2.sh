#!/bin/bash
# round 3, GPU call 2: full GPU suite (teacher-forced parity incl.), per-level HOG times and PMC attribution of the packed kernel
# (specialised / generic instance, ablations without folds / finish / column RMW / image loads)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3_run2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2; grep -E "^FAILED|teacher-forced" $O/pytest_gpu.txt | head -12
export SDM_HOG_MODES=2
python scripts/gpu_hogtime.py 2>&1 | grep mode > $O/hogtime.txt
SDM_HOG_NO_SPECIALISE=1 python scripts/gpu_hogtime.py 2>&1 | grep mode | sed 's/default/generic/' >> $O/hogtime.txt
for v in abl1 abl2 abl3 abl4 w8; do SDM_HIP_LIB=$R/exp/libsdm_$v.so python scripts/gpu_hogtime.py 2>&1 | grep mode >> $O/hogtime.txt; done
cat $O/hogtime.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU"
run_pmc() {  # tag lib env
  for i in 1 2 3; do
    eval pmc=\$P$i
    env $3 SDM_HIP_LIB=$2 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$1/p$i -o p -- python $R/scripts/gpu_hogtime.py > /dev/null 2> $O/pmc_$1_p$i.err
  done
  python $R/scripts/pmc_by_grid.py $O/pmc_$1 hog_packed > $O/pmc_$1.txt 2>&1
}
D=$R/superviseddescent_amd/lib/libsdm_hip.so
run_pmc spec $D SDM_X=0
run_pmc generic $D SDM_HOG_NO_SPECIALISE=1
for v in abl1 abl2 abl3; do run_pmc $v $R/exp/libsdm_$v.so SDM_X=0; done
rm -rf $O/pmc_*/p*/  # raw csv is large; the per-grid summaries stay
head -50 $O/pmc_spec.txt
