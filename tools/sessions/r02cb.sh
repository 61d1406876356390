#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02cb; mkdir -p $O
timeout 60 python tools/sessions/r02br.py 2>&1 | grep "^it\|fault" | cut -c1-200 | tee $O/diag.txt
grep -q fault $O/diag.txt && exit 1
timeout 900 python -m pytest tests/test_gpu_wride.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
BENCH_NOCPU=1 BENCH_ONLY="PTB" timeout 600 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-170
timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-200
