#!/bin/bash
# rocprofv3 recipes for the bench (run on the GPU box through gpurun).  Usage: scripts/profile.sh <tag>
# 1) kernel trace + stats, 2) separate PMC passes (never combined with sys/hip traces).
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --train-rows 640"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace_stdout.log 2>&1
# summaries
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pmc -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_${name}_stdout.log 2>&1
done
python - <<PY
import csv, glob, collections, os
out="$OUT"
rows=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]
        rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        cnt[(k,r["Counter_Name"])]+=1
with open(out+"/pmc_summary.txt","w") as fh:
    for k,v in rows.items():
        fh.write(k+"\n")
        for c,val in sorted(v.items()):
            n=cnt[(k,c)]
            fh.write(f"   {c:36s} total {val:.6g}  per-dispatch {val/n:.6g}  (n={n})\n")
print(open(out+"/pmc_summary.txt").read())
PY
echo ---- kernel stats; head -20 $OUT/kernel_stats.csv
