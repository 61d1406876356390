#!/bin/bash
# r03y5: rec3_fwd_kernel (forward when the batch has more rows than CUs): its storer wave at priority 3 (libA) against shipped (libC), B = 512 and 1024
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in A C; do
  for gb in 512 1024; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v B$gb', j['ms_per_step'], j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'])
"
  done
done
done
