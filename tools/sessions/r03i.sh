#!/bin/bash
# r03i: wgrad_mfma_kernel with (slot block, gate) tasks in mode 1
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print({k:j[k] for k in j if k in ('config','shape','ms_hipgraph','ms_eager','ms_per_step')})
"
