#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L" | tee gpurun_out/af_bench_stack.log
