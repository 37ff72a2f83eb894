#!/bin/bash
# PMC passes only (rocprofv3, csv output) for the bench; prints per-kernel per-dispatch averages.
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --train-rows 640"
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p${i}_stdout.log 2>&1
done
python - <<PY
import csv, glob, collections
out="$OUT"
rows=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-48:]
        rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
with open(out+"/pmc_summary.txt","w") as fh:
    for k,v in sorted(rows.items()):
        fh.write(k+"\n")
        for c,val in sorted(v.items()):
            fh.write(f"   {c:28s} per-dispatch {val/cnt[(k,c)]:.6g}  (n={cnt[(k,c)]})\n")
print(open(out+"/pmc_summary.txt").read())
PY
