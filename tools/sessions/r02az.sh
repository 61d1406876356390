#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BENCH_TS=1 VMLMF_STACK=0 timeout 300 python tools/bench_stack.py 180 16 64 9 2>&1 | grep "^L 1" | tee gpurun_out/az_prologue.log
BENCH_TS=1 VMLMF_STACK=0 timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L 1" | tee -a gpurun_out/az_prologue.log
BENCH_TS=1 VMLMF_STACK=1 timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L 1" | tee -a gpurun_out/az_prologue.log
