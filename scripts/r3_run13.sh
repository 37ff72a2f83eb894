#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run13; mkdir -p $O
for v in 0 1 0 1; do echo "SDM_GRAM_W16=$v"; SDM_GRAM_W16=$v python scripts/gram_timing.py 100000 2>&1 | tail -n 1; done | tee $O/gram_w16.txt
