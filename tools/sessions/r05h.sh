#!/bin/bash
# r05h: Adam in one launch with one verdict per step: tests + bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wride.py tests/test_gpu_modules.py -x -q 2>&1 | grep -E "^E |FAILED|passed|failed" | head
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extra > $O/new.json 2> $O/new.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05h/new.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), "train", d.get("train_step_ms"), "adam", d["adam_ms"], d["fused_adam_ms"], "eager", d["eager_ms_per_step"], {k:v for k,v in d["kernels_us"].items() if v})
PY
done
