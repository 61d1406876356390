#!/bin/bash
# r04h: rec4_bwd_kernel with two rows per workgroup (more rows than CUs): parity, then B = 512 / 384 / 300 with one and two rows per workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_inrow.py tests/test_gpu_wride.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels_us']; print('$1', j['config']['batch_per_gpu'], 'ms', j['ms_per_step'], 'train', j.get('train_step_ms'), 'fwd', k['rec_fwd_kernel'], 'bwd', k['rec_bwd_kernel'], 'wgrad', k['wgrad_mfma_kernel'], 'reduce', k['reduce_cg_kernel'])
"; }
for rep in 1 2; do
for b in 512 384 300; do
  for r in 1 2; do
    VMLMF_INROW_ROWS=$r timeout 600 python bench.py --global-batch $b --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>>$O/err.txt | line "rows=$r"
  done
done
done
timeout 600 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | line "A"
