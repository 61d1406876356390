#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q 2>&1 | grep -E "passed|failed|FAILED"
for v in 1 0; do echo "XWAVE=$v"; VMLMF_XWAVE=$v BENCH_NOCPU=1 BENCH_ONLY="A-group" timeout 200 python tools/bench_configs.py 2>&1 | grep config | cut -c1-150; VMLMF_XWAVE=$v BENCH_NOCPU=1 BENCH_ONLY="A/B" timeout 200 python tools/bench_configs.py 2>&1 | grep config | cut -c1-150; done
