#!/bin/bash
# bring-up of the wavefront forward (backward still chained per layer)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stack.py -x -q 2>&1 | tail -30 > gpurun_out/aa_stack.log
cat gpurun_out/aa_stack.log
