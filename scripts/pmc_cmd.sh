#!/bin/bash
# one PMC pass for an arbitrary command: scripts/pmc_cmd.sh <tag> "<counters>" <command...>; prints per-kernel averages + duration
set -u
TAG=$1; PMC=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -o p -- "$@" > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob, collections, re
rows=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel)', r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
        rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        dur[(k,r["Counter_Name"])]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k,v in sorted(rows.items()):
    print(k)
    for c,val in sorted(v.items()):
        n=cnt[(k,c)]
        print("   %-26s %14.6g per dispatch (n=%d)  avg duration %.1f us" % (c, val/n, n, dur[(k,c)]/n/1e3))
PY
