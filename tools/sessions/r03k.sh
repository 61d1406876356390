#!/bin/bash
# r03k: riding threshold re-swept with the faster backward rows (B = 96 .. 192), kernel breakdown at B = 128 / 256
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 96 112 128 160 192; do for mb in 96 256; do echo -n "B=$gb VMLMF_WRIDE_MAXB=$mb: "; VMLMF_WRIDE_MAXB=$mb timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['ms_per_step'], {a:b for a,b in k.items() if b>0}, d['loss'])"; done; done
