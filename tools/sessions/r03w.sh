#!/bin/bash
# r03w: s_setprio 3 in the mover waves - libA: rec_fwd storer + rec3 / rec_bwd movers; libB: A + the wavefront kernels' loader and storer;
# libD: B + the x-teams at priority 2; libC: before.  Same box, interleaved: config A (bench.py) and the other configs.
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in A B D C; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v A', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'])
"
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v', j['config'][:28], j.get('ms_hipgraph'))
"
done
done
