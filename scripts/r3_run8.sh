#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3_run8; mkdir -p $O
export SDM_HOG_MODES=2
python scripts/gpu_hogtime.py 2>&1 | grep mode > $O/hogtime.txt
for v in late tab ah4 latetab all; do SDM_HIP_LIB=$R/exp/libsdm_$v.so python scripts/gpu_hogtime.py 2>&1 | grep mode | sed "s#.*libsdm_#$v #" >> $O/hogtime.txt; done
python scripts/gpu_hogtime.py 2>&1 | grep mode >> $O/hogtime.txt
cat $O/hogtime.txt
for v in late all; do echo "== tests with $v"; SDM_HIP_LIB=$R/exp/libsdm_$v.so timeout 600 python -m pytest tests/test_gpu_packing.py -m gpu -q -x 2>&1 | tail -n 2; done
