#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3_run5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  SDM_BACKSOLVE_STEPS=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$mode -o t -- python $R/scripts/solve_only.py 8801 44 > $O/solve_$mode.txt 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/t$mode/t_kernel_stats.csv")))
print("mode $mode")
for r in rows[:12]:
    print("  %-60s calls %5s total %9.1f us avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
  tail -2 $O/solve_$mode.txt
done
rm -rf $O/t0/*_kernel_trace.csv $O/t1/*_kernel_trace.csv
