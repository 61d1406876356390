#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stack.py -x -q 2>&1 | tail -5
for m in 1; do echo "STACK=$m"; VMLMF_STACK=$m timeout 120 python tools/run_c_timing.py 2>&1 | grep WMIN; done
