#!/bin/bash
# r04w: epoch words zeroed by rb_pack_kernel / rb_zero_pad_kernel instead of memset nodes: parity, layer and LM times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_wring.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
[ $rc = 0 ] || exit 1
for i in 1 2; do
timeout 300 python tools/run_e.py 2>/dev/null | tail -1
timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1
timeout 300 python tools/bench_lm.py --only-head 2>/dev/null | head -1 | cut -c1-220
done
