#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q 2>&1 | grep -E "passed|failed|FAILED"
BENCH_ONLY=PTB BENCH_NOCPU=1 timeout 300 python tools/bench_configs.py 2>&1 | grep "PTB" | cut -c1-170 | tee gpurun_out/bh_e.log
timeout 300 python tools/bench_rb.py e32 2>&1 | grep shape | cut -c1-150 | tee -a gpurun_out/bh_e.log
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bh_prof -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph > /dev/null 2>&1 < /dev/null
