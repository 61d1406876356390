#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 0 1; do echo "STACK=$m"; VMLMF_STACK=$m timeout 300 python tools/run_c_timing.py; done > gpurun_out/ab_timing.log 2>&1
cd /tmp
VMLMF_STACK=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab_prof -o c -- python3 $GRAFT_REPO_ROOT/tools/run_c.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/ab_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls gpurun_out/ab_prof/*kernel_stats.csv | head -1)
head -14 $f | cut -c1-150 >> gpurun_out/ab_timing.log
cat gpurun_out/ab_timing.log
