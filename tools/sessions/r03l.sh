#!/bin/bash
# r03l: riding on / off at B = 32 .. 96 with rec3_bwd_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 32 64 72 80 88 96; do for w in 1 0; do echo -n "B=$gb VMLMF_WRIDE=$w: "; VMLMF_WRIDE=$w timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['ms_per_step'], k['rec_bwd_kernel'], k['wgrad_mfma_kernel'], d['loss'])"; done; done
