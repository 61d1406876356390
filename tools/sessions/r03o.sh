#!/bin/bash
# r03o: cluster gather of the row-block kernels through the XCD's own L2 (VMLMF_RB_NEAR=1) against system-scope loads (=0):
# config E layer time, per-kernel split, parity of the row-block tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03o
for near in 0 1; do
  echo "== VMLMF_RB_NEAR=$near"
  VMLMF_RB_NEAR=$near timeout 300 python tools/run_e.py 2>&1 | tail -2
  VMLMF_RB_NEAR=$near timeout 300 python tools/run_e.py --v3 2>&1 | tail -2
done
timeout 1500 python -m pytest tests/test_gpu_rb.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r03o/prof" -o e -- python3 "$GRAFT_REPO_ROOT/tools/run_e.py" --nograph > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/rocprof_summary.py gpurun_out/r03o/prof 2>/dev/null | head -12 || ls gpurun_out/r03o/prof
