#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 120 scripts/ubench/bin/chain_stamps 70 ) > gpurun_out/r5_chain_stamps_potrf4.log 2>&1
tail -22 gpurun_out/r5_chain_stamps_potrf4.log
