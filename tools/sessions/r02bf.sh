#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED"
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','eager_ms_per_step')}, d['roofline']['launch_us'])"; done
VMLMF_STACK=1 timeout 120 python tools/run_c_timing.py 2>&1 | grep WMIN
