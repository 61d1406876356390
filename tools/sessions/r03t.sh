#!/bin/bash
# r03t: wgrad_quad_kernel (four A tiles per wave for the dpre products) against wgrad_mfma_kernel (VMLMF_WGRAD_QUAD=0): parity on
# the tests that take the stand-alone weight-gradient kernel, then timing at B = 128 / 256 / 512 (config D legs) and the configs
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for q in 1 0; do
  echo "== VMLMF_WGRAD_QUAD=$q"
  for gb in 512 256 128; do VMLMF_WGRAD_QUAD=$q timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print($gb, j['ms_per_step'], j['kernels_us']['wgrad_mfma_kernel'], j['kernels_us']['reduce_cg_kernel'])
"; done
done
