#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
VMLMF_STACK=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ak_prof -o c -- python3 $GRAFT_REPO_ROOT/tools/run_c.py > /dev/null 2>&1 < /dev/null
