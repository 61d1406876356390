#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stack.py -x -q 2>&1 | tail -30 > gpurun_out/ad_stack.log
cat gpurun_out/ad_stack.log
for m in 0 1; do echo "STACK=$m"; VMLMF_STACK=$m timeout 120 python tools/run_c_timing.py 2>&1 | grep WMIN; done > gpurun_out/ad_timing.log 2>&1
cat gpurun_out/ad_timing.log
timeout 60 python tools/run_stack_probe.py 3
