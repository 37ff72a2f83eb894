#!/bin/bash
# Round 6, VERDICT r05 item 4: variants of the pixel kernel's LDS layout / accesses, built as exp/libsdm_hog_<name>.so from a COPY
# of csrc/sdm_hog_fast.hip with the ablation switches (scripts/experiments/hog_packed_ablations.patch) applied -- the shipped
# source carries none.
#   base      as shipped (column-row stride 66)
#   abl3      no per-pixel column read-modify-write (HP_ABL=3)
#   st64      column-row stride 64: the read-modify-write's bank depends on the lane only (conflict-free); the fold's operand reads fall into ONE bank
#   abl15     stride 66, fold operands read conflict-free from the lane's own column (wrong products: what a perfect operand layout could save)
#   st64abl15 both: no bank conflict left anywhere in the row loop or the folds
# then: scripts/r6_hog_lds_run.sh on the GPU box (results: profiles/r06_hog_lds.txt).  Other HP_ABL modes: pass -DHP_ABL=n yourself.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC=$ROOT/superviseddescent_amd/csrc
make -s -C $CSRC >/dev/null
mkdir -p $ROOT/exp/src
cp $CSRC/*.hip $CSRC/*.h $CSRC/*.inc $ROOT/exp/src/
(cd $ROOT/exp/src && patch -s -p3 < $ROOT/scripts/experiments/hog_packed_ablations.patch)
build() {   # name, flags
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include "$@" -c $ROOT/exp/src/sdm_hog_fast.hip -o $ROOT/exp/hog_fast_$name.o
  objs=""
  for o in $ROOT/superviseddescent_amd/lib/obj/*.o; do case $o in */sdm_hog_fast.o) ;; *) objs="$objs $o";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/exp/libsdm_hog_$name.so $objs $ROOT/exp/hog_fast_$name.o -ldl
  echo built exp/libsdm_hog_$name.so
}
build base
build abl3 -DHP_ABL=3
build st64 -DHP_ST=64
build abl15 -DHP_ABL=15
build st64abl15 -DHP_ST=64 -DHP_ABL=15
