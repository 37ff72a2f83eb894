#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/superviseddescent_amd/lib
( timeout 900 python scripts/r5_detect_env_ab.py SDM_HIP_LIB $L/libsdm_hip_old.so,$L/libsdm_hip.so,$L/libsdm_hip_old.so,$L/libsdm_hip.so 4096 40 ) > gpurun_out/r5_reduce_ab.log 2>&1
tail -6 gpurun_out/r5_reduce_ab.log | cut -c1-300
