#!/bin/bash
# r03s: rec_fwd_kernel with TWO storer waves beside the x-projection wave (gpurun_in/libA.so) against the library before
# (libC.so), same box, interleaved; then the forward parity tests on the new library
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in A C; do
  VMLMF_LIB="$GRAFT_REPO_ROOT/gpurun_in/lib$v.so" timeout 600 python bench.py --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$rep lib$v', j['ms_per_step'], j.get('ms_per_step_kept_images'), j.get('train_step_ms'), j['kernels_us']['rec_fwd_kernel'], j['kernels_us']['rec_bwd_kernel'], 'loss', j['loss'])
"
done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_wride.py -x -q -m gpu 2>&1 | tail -3
