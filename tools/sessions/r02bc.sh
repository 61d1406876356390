#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/run_stack_l1.py 256 77 24 1024,2048 24 2 2>&1 | grep "^L" | tee gpurun_out/bc_batch.log
timeout 600 python tools/run_stack_l1.py 180 77 16 128,256,512 24 1 2>&1 | grep "^L" | tee -a gpurun_out/bc_batch.log
