#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/superviseddescent_amd/lib
for v in old new old new; do
  if [ $v = old ]; then export SDM_HIP_LIB=$L/libsdm_hip_old.so; else export SDM_HIP_LIB=$L/libsdm_hip.so; fi
  echo "== $v" >> gpurun_out/r5_fold_ab.log
  ( timeout 600 python scripts/r5_rcr68_train_probe.py 100000 22 ) 2>/dev/null | tail -1 >> gpurun_out/r5_fold_ab.log
done
for v in old new; do
  if [ $v = old ]; then export SDM_HIP_LIB=$L/libsdm_hip_old.so; else export SDM_HIP_LIB=$L/libsdm_hip.so; fi
  echo "== $v" >> gpurun_out/r5_fold_ab.log
  ( timeout 600 python scripts/r5_rcr68_train_probe.py 100000 68 ) 2>/dev/null | tail -1 >> gpurun_out/r5_fold_ab.log
done
cat gpurun_out/r5_fold_ab.log | cut -c1-400
