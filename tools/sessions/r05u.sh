#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 512 384; do for xw in 1 0; do
VMLMF_XWAVE=$xw timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_us']; print($gb, 'xwave=$xw', d['ms_per_step'], d['train_step_ms'], {a:b for a,b in k.items() if b>0})"; done; done
