#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fine_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fine_trace -o t -- python $R/scripts/r5_solve_ab.py --child 8801 44 4096 /tmp/x.npy > $R/gpurun_out/fine_trace.log 2>&1
python - <<PY
import csv,glob,re,collections
f=glob.glob('$R/gpurun_out/fine_trace/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n):
    m=re.search(r'(\w+_kernel)', n); return m.group(1) if m else n[:30]
idx=[i for i,r in enumerate(rows) if 'diag_absmax_kernel' in r['Kernel_Name']]
seg=rows[idx[-1]:]
t0=int(seg[0]['Start_Timestamp'])
with open('$R/gpurun_out/trace22_timeline.txt','w') as fh:
    for r in seg:
        fh.write("%9.1f %8.1f %-34s grid=%s q=%s\n" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, short(r['Kernel_Name']), r['Grid_Size_X'], r.get('Queue_Id')))
PY
rm -rf $R/gpurun_out/fine_trace
head -48 $R/gpurun_out/trace22_timeline.txt; tail -8 $R/gpurun_out/trace22_timeline.txt
