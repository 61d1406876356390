#!/bin/bash
# r03q: same-box A/B of three builds of the library (gpurun_in/lib{A,B,C}.so: criterion tail as a function of its own / inlined
# into head_epilogue / the library before the riding criterion), interleaved twice: box-to-box variance is ~3 %, larger than the effect
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03q
for rep in 1 2; do
for v in A C; do
  for mode in riding separate; do
    [ "$v" = C ] && [ "$mode" = riding ] && continue
    extra=""; [ "$mode" = separate ] && extra="--separate-loss"
    VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --steps 300 --warmup 30 --no-extra --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v $mode', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['kernels_us']['rec_fwd_kernel'])
"
  done
done
done
