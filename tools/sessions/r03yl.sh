#!/bin/bash
# r03yl: pack_kernel with 32 dot-product elements per workgroup instead of 8 (libA) against libC
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for v in A C; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v A', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'])
"
done
done
