#!/bin/bash
# loop-alignment sweep of the headline kernels' translation units (rec_fwd_kh16, rec3, pack): same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05p; mkdir -p $O
for rep in 1 2 3; do
  for v in base al32 al64 al128; do
    L=$GRAFT_REPO_ROOT/.abtree/$v/libvmlmf_hip.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so
    VMLMF_LIB=$L timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels_us']
print('$v', $rep, d['ms_per_step'], d['train_step_ms'], k['rec_fwd_kernel'], k['rec_bwd_kernel'], k['finish2_kernel'])"
  done
done | tee $O/sweep.txt
