#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3_run9; mkdir -p $O
export SDM_HOG_MODES=2
python scripts/gpu_hogtime.py 2>&1 | grep mode > $O/hogtime.txt
for v in stag16 stag64 late; do SDM_HIP_LIB=$R/exp/libsdm_$v.so python scripts/gpu_hogtime.py 2>&1 | grep mode | sed "s#.*libsdm_#$v #" >> $O/hogtime.txt; done
python scripts/gpu_hogtime.py 2>&1 | grep mode >> $O/hogtime.txt
cat $O/hogtime.txt
