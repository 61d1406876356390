#!/bin/bash
# r04k: wgrad_ring_kernel (weight gradients of large layers through an LDS ring): parity, then config E layer / plain layer / LM step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wring.py -x -q -m gpu > $O/tests_wring.txt 2>&1; echo "wring tests rc=$?"; grep -E "passed|failed" $O/tests_wring.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests_wring.txt | head -30
for rep in 1 2; do
for m in 0 -1; do
  echo "== VMLMF_WRING=$m"
  VMLMF_WRING=$m timeout 300 python tools/run_e.py 2>/dev/null | tail -1
  VMLMF_WRING=$m timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1
  VMLMF_WRING=$m timeout 300 python tools/bench_lm.py 2>/dev/null | head -1 | cut -c1-220
done
done
cd /tmp && export TMPDIR=/tmp
VMLMF_WRING=-1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_e -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof_e -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
timeout 1500 python -m pytest tests/test_gpu_rb.py tests/test_gpu_modules.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -30
