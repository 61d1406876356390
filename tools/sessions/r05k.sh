#!/bin/bash
# r05k: bf16 gate tape on the wavefront kernels (dtype bf16 below 4096 rows): parity tests + config C timing
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_stack.py -x -q -s 2>&1 | grep -E "^E |FAILED|passed|failed|config C bf16" | head -40
python - <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench
r = bench.other_configs(iters=200)
print({k: (v.get("ms_per_step"), v.get("vs_fp32"), v.get("error")) for k, v in r.items()})
PY
