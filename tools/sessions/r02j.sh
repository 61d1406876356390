#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], 'eager', d['eager_ms_per_step'], {k:v for k,v in d['kernels_us'].items() if v})"; done
