#!/bin/bash
# kernel trace of the factor + solve chain at F = 27 201 (random system), float16 and f32 trailing updates
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  SDM_UPDATE_F32=$m rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/solve_tr$m -o t -- python $REPO/scripts/update_f16_ab.py --child 27201 136 4096 /tmp/x.npy > /dev/null 2>&1
  echo "SDM_UPDATE_F32=$m"; python - <<PY
import csv
for r in list(csv.DictReader(open("$REPO/gpurun_out/solve_tr$m/t_kernel_stats.csv")))[:9]:
    print("%-50s %6s %10.1f us avg %10.1f ms total" % (r["Name"].replace("(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
rm -rf $REPO/gpurun_out/solve_tr*/t_kernel_trace.csv
