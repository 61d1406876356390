#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/run_stack_l1.py 256 77 24 128,192,256,384,512 24 2 2>&1 | grep "^L" | tee gpurun_out/bb_batch.log
timeout 600 python tools/run_stack_l1.py 180 9 16 64,128,192,256 128 2 2>&1 | grep "^L" | tee -a gpurun_out/bb_batch.log
