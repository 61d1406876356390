#!/bin/bash
# r03x: libA = rec_fwd storer at priority 3 (kept); libB = A + rec3_bwd's COMPUTE waves at priority 3; libD = storer at priority 1 instead of 3;
# libC = before.  Same box, interleaved, config A.
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for v in A B D C; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v A', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'])
"
done
done
