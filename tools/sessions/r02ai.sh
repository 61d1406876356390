#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in 1 2 3; do echo "ABL $a"; WF_ABL=$a timeout 300 python tools/bench_stack.py 256 24 64 77 2>&1 | grep "^L 1"; done | tee gpurun_out/ai_abl.log
