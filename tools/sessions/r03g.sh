#!/bin/bash
# r03g: reduce_cg + finish in one launch for the riding layers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wride.py tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -4
for v in 1 0; do echo -n "fused=$v: "; VMLMF_FUSED_FINISH=$v timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_kept_images'], d['train_step_ms'], d['eager_ms_per_step'], d['kernels_us'], d['loss'])"; done
