#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/run_stack_l1.py 2>&1 | grep "^L" | tee gpurun_out/ba_l1.log
