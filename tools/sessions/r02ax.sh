#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stack.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L [12]" | tee gpurun_out/ax_bench_stack.log
VMLMF_STACK=1 timeout 120 python tools/run_c_timing.py 2>&1 | grep WMIN
