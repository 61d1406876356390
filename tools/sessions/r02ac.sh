#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 python tools/run_stack_probe.py 3 > gpurun_out/ac_plain.log 2>&1
cd /tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ac_prof -o c -- python3 $GRAFT_REPO_ROOT/tools/run_stack_probe.py 3 > $GRAFT_REPO_ROOT/gpurun_out/ac_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/ac_plain.log; tail -20 gpurun_out/ac_prof.log
