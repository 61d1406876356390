#!/bin/bash
# r03j: rec3_fwd_kernel (x side in the compute waves, ~170 VGPRs: two workgroups per CU) against rec_fwd_kernel's x-projection wave (256
# VGPRs: one per CU) at batches above 64 on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
for gb in 64 128 256 512; do for r in 2 3; do echo -n "B=$gb VMLMF_REC3=$r: "; VMLMF_REC3=$r timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_kept_images'], d['kernels_us']['rec_fwd_kernel'], d['kernels_us']['rec_bwd_kernel'], d['loss'])"; done; done
