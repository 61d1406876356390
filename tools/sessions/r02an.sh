#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/an_gpu_tests.log
for m in auto 0; do for c in "demo.sh: OPP V1" "C(fp32)"; do echo "STACK=$m"; VMLMF_STACK=$m BENCH_NOCPU=1 BENCH_ONLY="$c" timeout 200 python tools/bench_configs.py 2>&1 | grep config; done; done | tee gpurun_out/an_configs.log
