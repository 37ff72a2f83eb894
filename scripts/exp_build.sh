#!/bin/bash
# builds experiment variants of libsdm_hip.so: exp/libsdm_expN.so with -DSDM_EXP=N on sdm_hog_fast.hip
set -e
cd "$(dirname "$0")/../superviseddescent_amd/csrc"
mkdir -p ../../exp
make >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSDM_EXP=$n -c sdm_hog_fast.hip -o ../../exp/hog_fast_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/libsdm_exp$n.so ../lib/obj/sdm_hog.o ../../exp/hog_fast_$n.o ../lib/obj/sdm_apply.o ../lib/obj/sdm_solve.o ../lib/obj/sdm_capi.o
done
