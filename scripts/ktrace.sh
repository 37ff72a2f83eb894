#!/bin/bash
# kernel trace + per-kernel stats for an arbitrary command:  scripts/ktrace.sh <tag> <command...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/ktrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- "$@" > $OUT/stdout.log 2>&1
tail -3 $OUT/stdout.log
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print("%-70s %8s %12s %10s %6s" % ("kernel","calls","total_us","avg_us","%"))
    for r in rows[:25]:
        print("%-70s %8s %12.1f %10.2f %6.2f" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
