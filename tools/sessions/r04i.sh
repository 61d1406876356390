#!/bin/bash
# r04i: the health-word form of the optimizer guard: the give-up tests, the stress tool, train_step_ms
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wride.py tests/test_gpu_modules.py tests/test_gpu_inrow.py tests/test_gpu_stack.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
for m in 1; do
VMLMF_ADAM_GUARD=$m timeout 600 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('guard=$m', j['ms_per_step'], j.get('ms_per_step_kept_images'), 'train', j.get('train_step_ms'), 'fused_adam', j['fused_adam_ms'])
"
done
