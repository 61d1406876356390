#!/bin/bash
# r03y9: dpre stores of the backward (rec3_bwd without riding workers: B > 64; wf_bwd) with the nt hint: libA against libC
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in A C; do
  for gb in 512; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v B$gb', j['ms_per_step'], j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'])
"
  done
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l)
        if j['config'][:2] in ('C(', 'de', 'D@'): print('$rep lib$v', j['config'][:28], j.get('ms_hipgraph'))
"
done
done
