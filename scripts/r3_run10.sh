#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run10; mkdir -p $O
./scripts/ubench/bin/valu_rates > $O/valu_rates.txt 2>&1; grep -E "CND|CMP" $O/valu_rates.txt
python scripts/gpu_latency.py 4096 2>&1 | tee $O/latency.txt
