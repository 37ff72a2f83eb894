#!/bin/bash
# round 5, GPU call 12: row-update kernel with two buffers and the C prefetch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for F in "8801 44" "17051 44" "27201 136"; do
  set -- $F
  ( timeout 600 python scripts/r5_solve_ab.py $1 $2 4096 4,0 ) > gpurun_out/r5_solve_ab_$1_thin2.log 2>&1
done
( timeout 1500 python -m pytest tests/test_gpu_sharded_solve.py tests/test_gpu_solver_accuracy.py tests/test_gpu_exchange.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r5_run12_tests.log 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5_trace_solve -o t -- python $GRAFT_REPO_ROOT/scripts/solve_only.py 8801 44 > $GRAFT_REPO_ROOT/gpurun_out/r5_trace_solve.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
geo=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r5_trace_solve/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        geo[r['Kernel_Name'].split('(')[0][-40:]].append((float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3)
for k,v in sorted(geo.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print("%-42s calls %4d total %9.1f avg %7.2f" % (k, len(v), sum(v), sum(v)/len(v)))
PY
tail -n 1 gpurun_out/r5_solve_ab_*_thin2.log; tail -5 gpurun_out/r5_run12_tests.log
