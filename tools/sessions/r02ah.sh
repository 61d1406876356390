#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VMLMF_STACK=0 timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L [12]" | tee gpurun_out/ah_bench_chain.log
