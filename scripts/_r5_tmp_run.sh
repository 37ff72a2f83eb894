#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/solve_timeline.sh 8801 44 $PWD/gpurun_out/r05_solve_timeline_rcr22.txt
bash scripts/solve_timeline.sh 27201 136 $PWD/gpurun_out/r05_solve_timeline_rcr68.txt
wc -l gpurun_out/r05_solve_timeline_*.txt; tail -4 gpurun_out/r05_solve_timeline_rcr68.txt
