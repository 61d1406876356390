#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in 1 2; do echo "SKINNY=$v"; VMLMF_SKINNY=$v BENCH_ONLY=PTB BENCH_NOCPU=1 timeout 300 python tools/bench_configs.py 2>&1 | grep "PTB"; done | tee gpurun_out/at_e_skinny.log
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/at_prof -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py > /dev/null 2>&1 < /dev/null
