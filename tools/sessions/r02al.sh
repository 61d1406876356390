#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stack.py -x -q 2>&1 | tail -3
VMLMF_STACK=1 timeout 120 python tools/run_c_timing.py 2>&1 | grep WMIN
cd /tmp
VMLMF_STACK=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/al_prof -o c -- python3 $GRAFT_REPO_ROOT/tools/run_c.py > /dev/null 2>&1 < /dev/null
