#!/bin/bash
# HBM traffic + MFMA busy of the Gram launch alone (scripts/gram_timing.py at 100 000 rows): rocprofv3 PMC passes, one counter
# set per pass.  usage: scripts/gram_pmc.sh <tag> [path of libsdm_hip.so to test]
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -n "${2:-}" ] && export SDM_HIP_LIB=$2
OUT=$REPO/gpurun_out/gram_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python $REPO/scripts/gram_timing.py ${GRAM_ARGS:-} > /dev/null 2> $OUT/p${i}_stderr.log
done
python $REPO/scripts/pmc_by_grid.py $OUT syrk > $OUT/summary.txt
cat $OUT/summary.txt
