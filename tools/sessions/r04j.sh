#!/bin/bash
# r04j: weight gradients inside the clustered backward (rb_bwd_kernel<WG>, one-group rank <= 32 layers): parity, then the plain PTB layer and the LM step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rb.py tests/test_gpu_modules.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -30
for rep in 1 2; do
for m in 1 0; do
  echo "== VMLMF_RB_WGRAD=$m"
  VMLMF_RB_WGRAD=$m timeout 300 python tools/run_e.py --v3 --nograph 2>/dev/null | tail -1
  VMLMF_RB_WGRAD=$m timeout 300 python tools/bench_lm.py 2>/dev/null | head -1 | cut -c1-220
  VMLMF_RB_WGRAD=$m timeout 300 python tools/bench_lm.py 32 2>/dev/null | head -1 | cut -c1-220
done
done
