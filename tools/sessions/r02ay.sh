#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 2 1; do for c in 64 128 32; do echo "WNS=$w WCHUNKS=$c"; VMLMF_WNS=$w VMLMF_WCHUNKS=$c timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step',)})"; done; done | tee gpurun_out/ay_wns.log
