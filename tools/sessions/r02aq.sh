#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 64 32 16; do echo "WCHUNKS=$w"; VMLMF_WCHUNKS=$w BENCH_ONLY="E: PTB V4" timeout 300 python tools/bench_configs.py 2>&1 | grep config; done | tee gpurun_out/aq_e_wchunks.log
