#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in auto 1; do for c in "demo.sh: OPP V1" "C(fp32)"; do echo "STACK=$m"; VMLMF_STACK=$m BENCH_NOCPU=1 BENCH_ONLY="$c" timeout 200 python tools/bench_configs.py 2>&1 | grep config; done; done | tee gpurun_out/am_configs.log
