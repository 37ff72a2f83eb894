#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3_run7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for fuse in 1 0; do
  SDM_SOLVE_NO_FUSE=$fuse rocprofv3 --kernel-trace --output-format csv -d $O/t$fuse -o t -- python $R/scripts/solve_only.py 8801 44 > $O/solve_$fuse.txt 2>&1
  tail -1 $O/solve_$fuse.txt
done
ls -la $O/t0 $O/t1
