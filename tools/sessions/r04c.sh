#!/bin/bash
# r04c: rec4_bwd_kernel (weight gradients inside the rows' workgroups): parity, then A/B against the stand-alone / riding forms
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_inrow.py -x -q -m gpu > $O/tests.txt 2>&1; echo "inrow tests rc=$?"; tail -25 $O/tests.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels_us']; print('$1', j['config']['batch_per_gpu'], 'ms', j['ms_per_step'], 'kept', j.get('ms_per_step_kept_images'), 'train', j.get('train_step_ms'), 'fwd', k['rec_fwd_kernel'], 'bwd', k['rec_bwd_kernel'], 'wgrad', k['wgrad_mfma_kernel'], 'reduce', k['reduce_cg_kernel'], 'finish', k['finish_kernel'])
"; }
for rep in 1 2; do
for b in 64 128 256 512; do
  for m in 0 1; do
    VMLMF_INROW=$m timeout 600 python bench.py --global-batch $b --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>>$O/err.txt | tee -a $O/strong_inrow$m.jsonl | line "inrow=$m"
  done
done
done
