#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
for v in 1 0; do echo "WRIDE=$v"; VMLMF_WRIDE=$v BENCH_NOCPU=1 BENCH_ONLY="A/B" timeout 200 python tools/bench_configs.py 2>&1 | grep config | cut -c1-150; done
for k in 8 12 24 32; do echo "K=$k"; VMLMF_WRIDE_K=$k BENCH_NOCPU=1 BENCH_ONLY="A/B" timeout 200 python tools/bench_configs.py 2>&1 | grep config | cut -c60-150; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-400
