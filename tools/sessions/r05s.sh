#!/bin/bash
# finish2_kernel beside the backward launch (VMLMF_FINISH2=2) against behind it (=1): same box, interleaved; then the tests that cover it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05s; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_wride.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2 3; do
  for v in 1 2; do
    VMLMF_FINISH2=$v timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels_us']
print('finish2=$v', $rep, d['ms_per_step'], d['train_step_ms'], d['eager_ms_per_step'], k['rec_fwd_kernel'], k['rec_bwd_kernel'], k['finish2_kernel'])"
  done
done | tee $O/ab.txt
