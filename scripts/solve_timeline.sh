#!/bin/bash
# rocprofv3 kernel timeline of ONE factor + solve (the last of the four solves of scripts/r5_solve_ab.py --child F M rows):
# start (us), duration (us), kernel, grid, queue -- what DESIGN.md 4.5's account of the chain is read from.
#   scripts/solve_timeline.sh F M out.txt
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
R=$PWD; F=$1; M=$2; OUT=$3
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_trace -o t -- python $R/scripts/r5_solve_ab.py --child $F $M 4096 /tmp/x.npy > $R/gpurun_out/tl_trace.log 2>&1
python - <<PY
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob('$R/gpurun_out/tl_trace/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: (re.search(r'(\w+_kernel)', n).group(1) if re.search(r'(\w+_kernel)', n) else n[:34])
idx = [i for i, r in enumerate(rows) if 'diag_absmax_kernel' in r['Kernel_Name']]
seg = rows[idx[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
with open('$OUT', 'w') as fh:
    fh.write("# rocprofv3 --kernel-trace of one factor + solve, F = $F, $M right-hand sides (scripts/solve_timeline.sh): start us, duration us, kernel, grid (threads), queue\n")
    for r in seg:
        fh.write("%9.1f %8.1f %-34s grid=%s q=%s\n" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, short(r['Kernel_Name']), r['Grid_Size_X'], r.get('Queue_Id')))
PY
rm -rf $R/gpurun_out/tl_trace
