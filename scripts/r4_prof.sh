#!/bin/bash
# kernel trace of the detect step (split + fused), per-kernel averages
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
SDM_R4_ONLY=${1:-fused} rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r4_tr -o t -- python $REPO/scripts/r4_check_split.py 4096 t > $REPO/gpurun_out/r4_tr_stdout.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$REPO/gpurun_out/r4_tr/**/t_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s %6s %10.1f us avg %10.2f ms total" % (r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
rm -rf $REPO/gpurun_out/r4_tr/*/*kernel_trace.csv
