#!/bin/bash
# r04d: where does rec4_bwd_kernel lose its 18 us against rec3_bwd_kernel's rows?  ablation builds (libvmlmf_hip_exp.so, VMLMF_R4_ABL)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_inrow.py -x -q -m gpu > $O/tests.txt 2>&1; echo "inrow tests rc=$?"; tail -5 $O/tests.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels_us']; print('$1', j['config']['batch_per_gpu'], 'ms', j['ms_per_step'], 'fwd', k['rec_fwd_kernel'], 'bwd', k['rec_bwd_kernel'], 'wgrad', k['wgrad_mfma_kernel'], 'reduce', k['reduce_cg_kernel'])
"; }
export VMLMF_LIB="$GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip_exp.so" VMLMF_INROW=1
for rep in 1 2; do
for abl in 0 2 34 64 96; do
  VMLMF_R4_ABL=$abl timeout 600 python bench.py --global-batch 128 --steps 100 --warmup 10 --no-extra --no-cpu-baseline --no-graph 2>>$O/err.txt | line "abl=$abl"
done
done
