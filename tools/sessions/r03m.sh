#!/bin/bash
# r03m: wgrad_mfma_kernel batch size (row pairs per batch) for the one- and two-tile instantiations
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 128 256 512; do echo -n "B=$gb: "; timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['ms_per_step'], {a:b for a,b in k.items() if b>0}, d['loss'])"; done
timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print({k:j[k] for k in j if k in ('config','ms_hipgraph','ms_eager')})
"
