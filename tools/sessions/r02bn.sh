#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "$*"; env "$@" BENCH_NOCPU=1 BENCH_ONLY="A/B" timeout 200 python tools/bench_configs.py 2>&1 | grep config | cut -c60-150; }
run VMLMF_WRIDE=0
run VMLMF_WRIDE_DRY=1
run VMLMF_WRIDE_DRY=1 VMLMF_WRIDE_LAG=4
run VMLMF_WRIDE_DRY=1 VMLMF_WRIDE_LAG=8
run VMLMF_WRIDE_DRY=1 VMLMF_WRIDE_K=1
run VMLMF_WRIDE_LAG=4
run VMLMF_WRIDE_LAG=8
run VMLMF_WRIDE_LAG=8 VMLMF_WRIDE_K=8
