#!/bin/bash
# r03yb: the driver's invocation (--steps 20 --warmup 5): end-of-region wait by event polling + synchronize (VMLMF_BENCH_SPIN=1, default) against
# synchronize alone (=0); and --steps 300 for reference
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for spin in 1 0; do
  VMLMF_BENCH_SPIN=$spin timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep spin=$spin K20', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['eager_ms_per_step'])
"
done
done
VMLMF_BENCH_SPIN=1 timeout 600 python bench.py --gpus 1 --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('K300', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['eager_ms_per_step'])
"
