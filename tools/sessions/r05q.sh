#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05q
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05q/gpu_tier.txt
cat gpurun_out/r05q/gpu_tier.txt
for gb in 128 256 512; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print($gb, d['ms_per_step'], d['train_step_ms'])"; done | tee gpurun_out/r05q/strong.txt
