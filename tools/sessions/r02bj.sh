#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export VMLMF_SKINNY=0
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bj_prof -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph > /dev/null 2>&1 < /dev/null
