#!/bin/bash
# r04zb: the x side inside the clustered forward (VMLMF_RB_XFOLD=1) again, now that the step's loads sit in the exchange's shadow
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for m in 0 1; do
  echo "xfold=$m: group $(VMLMF_RB_XFOLD=$m timeout 300 python tools/run_e.py 2>/dev/null | tail -1 | cut -c1-40) | v3 $(VMLMF_RB_XFOLD=$m timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1 | cut -c1-40) | b32 $(VMLMF_RB_XFOLD=$m timeout 300 python tools/run_e.py --batch 32 2>/dev/null | tail -1 | cut -c1-40)"
done; done
timeout 300 python -m pytest tests/test_gpu_rb.py -x -q -m gpu -k "x_side" 2>&1 | tail -1
