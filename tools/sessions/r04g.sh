#!/bin/bash
# r04g: x-fold inside rb_fwd_kernel (clustered layers): parity (test_gpu_rb, fuzz with the row-block kernels), then config E and the LM step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rb.py tests/test_gpu_wride.py tests/test_gpu_modules.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
for rep in 1 2; do
for m in 1 0; do
  echo "== VMLMF_RB_XFOLD=$m"
  VMLMF_RB_XFOLD=$m timeout 300 python tools/run_e.py --nograph 2>/dev/null | tail -1
  VMLMF_RB_XFOLD=$m timeout 300 python tools/run_e.py --v3 --nograph 2>/dev/null | tail -1
  VMLMF_RB_XFOLD=$m timeout 300 python tools/bench_lm.py 2>/dev/null | head -2 | cut -c1-200
  VMLMF_RB_XFOLD=$m timeout 300 python tools/bench_lm.py 32 2>/dev/null | head -2 | cut -c1-200
done
done
timeout 600 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('A', j['ms_per_step'], j.get('ms_per_step_kept_images'), 'train', j.get('train_step_ms'), 'adam', j['adam_ms'], j['fused_adam_ms'])
"
