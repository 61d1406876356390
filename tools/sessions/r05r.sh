#!/bin/bash
# A/B on one box: library before / after a change (VMLMF_LIB), interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05r; mkdir -p $O
for rep in 1 2 3; do
  for v in pre new; do
    L=$GRAFT_REPO_ROOT/.abtree/$v/libvmlmf_hip.so; [ $v = new ] && L=$GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so
    VMLMF_LIB=$L timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels_us']
print('$v', $rep, d['ms_per_step'], d['train_step_ms'], k['rec_fwd_kernel'], k['rec_bwd_kernel'], k['finish2_kernel'])"
  done
done | tee $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_wride.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
