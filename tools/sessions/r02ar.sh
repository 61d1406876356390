#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for v in 1 0; do echo "XEXP=$v"; VMLMF_XEXP=$v BENCH_ONLY=PTB timeout 300 python tools/bench_configs.py 2>&1 | grep "PTB"; done | tee gpurun_out/ar_e_xexp.log
