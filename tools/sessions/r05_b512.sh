#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 512 384 256; do
timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_us']; print($gb, d['ms_per_step'], d['train_step_ms'], {a:b for a,b in k.items() if b>0})"; done
timeout 600 python -m pytest tests/test_gpu_inrow.py -m gpu -q -x 2>&1 | tail -2
