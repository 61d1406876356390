#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rb.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
for v in 1 0; do echo "RB_TAG=$v"; VMLMF_RB_TAG=$v BENCH_ONLY=PTB BENCH_NOCPU=1 timeout 300 python tools/bench_configs.py 2>&1 | grep "PTB" | cut -c1-170; VMLMF_RB_TAG=$v timeout 300 python tools/bench_rb.py e32 2>&1 | grep "cluster of 16" | cut -c1-130; done | tee gpurun_out/bi_tag.log
