#!/bin/bash
# kernel trace of the Gram launch: split pre-pass vs product kernel, 16-wave and 8-wave forms
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 1 5; do
  SDM_GRAM_WIDE=$w rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/gram_w$w -o t -- python $REPO/scripts/gram_timing.py 100000 > /dev/null 2>&1
  echo "WIDE=$w"; python - <<PY
import csv
for r in list(csv.DictReader(open("$REPO/gpurun_out/gram_w$w/t_kernel_stats.csv")))[:6]:
    print("%-60s %5s %10.1f us avg" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
