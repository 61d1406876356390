#!/bin/bash
# r03n: wf_bwd_kernel with the in-lane gate fold + DPP adds (rec3_bwd's reduce) in the compute waves: parity + config C timing
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests/test_gpu_stack.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print({k:j[k] for k in j if k in ('config','ms_hipgraph','ms_eager')})
"
timeout 300 python tools/bench_stack.py 2>/dev/null | tail -12
