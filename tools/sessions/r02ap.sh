#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_rb.py e 2>&1 | grep "cluster of 16" | tee gpurun_out/ap_e_sc1.log
timeout 600 python -m pytest tests/test_gpu_rb.py -x -q 2>&1 | tail -3
