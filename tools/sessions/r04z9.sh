#!/bin/bash
# r04z9: does PyTorch's TunableOp find faster library kernels for the LM head's three GEMMs (R 8960, H 650, V 10000, fp32)?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04z9; mkdir -p $O
cd $O
for rep in 1 2; do
echo "plain:   $(timeout 300 python ../../tools/bench_lm.py --only-head 2>/dev/null | head -1 | cut -c150-230)"
echo "tunable: $(PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0 timeout 900 python ../../tools/bench_lm.py --only-head 2>/dev/null | head -1 | cut -c150-230)"
done
ls; head -20 tunableop_results0.csv 2>/dev/null | cut -c1-200
