#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','eager_ms_per_step','train_step_ms')})"
BENCH_NOCPU=1 BENCH_ONLY="C(fp32)" timeout 300 python tools/bench_configs.py 2>&1 | grep config
