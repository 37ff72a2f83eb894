#!/bin/bash
# On the GPU box: time (scripts/r4_abl.py: pixel kernel, sum of the four levels at 4 096 faces) and LDS counters of every variant of r6_hog_lds_variants.sh
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r6_hog_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base abl3 st64 abl15 st64abl15 base; do
  export SDM_HIP_LIB=$REPO/exp/libsdm_hog_$v.so
  python $REPO/scripts/r4_abl.py 2>/dev/null | tail -1
done | tee $OUT/times.txt
for v in base abl3 st64 abl15 st64abl15; do
  export SDM_HIP_LIB=$REPO/exp/libsdm_hog_$v.so
  rm -rf $OUT/p_$v
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p_$v -o pmc -- python $REPO/scripts/r4_abl.py > /dev/null 2> $OUT/p_${v}_stderr.log
  echo "== $v"; python $REPO/scripts/pmc_by_grid.py $OUT/p_$v hog_packed
  rm -rf $OUT/p_$v
done | tee $OUT/pmc.txt
