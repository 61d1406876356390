#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02l; mkdir -p $O
python -m pytest tests/test_gpu_rb.py -q -x 2>&1 | tail -3
python tools/bench_rb.py e32 2>/dev/null | cut -c1-300
python - <<'PY'
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_rb as b
from vmlmf_amd import _lib
for B in (64, 128, 256):
    b.run(f"E: PTB V4 group H=650 [32,32] B={B} T=35 x2 layers (auto)", *b.lm(B, True), -1, 10)
PY
