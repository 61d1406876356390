#!/bin/bash
# r03h: wgrad2_kernel (B through LDS, A prefetched, gate-major mode-1 tasks): parity + timing at C and E
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for v in 1 0; do echo "== VMLMF_WGRAD2=$v"; VMLMF_WGRAD2=$v timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print({k:j[k] for k in j if k in ('config','shape','ms_hipgraph','ms_eager','ms_per_step')})
"; done
