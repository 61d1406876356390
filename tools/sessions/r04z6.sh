#!/bin/bash
# r04z6: after keeping the one-workgroup row-block kernels' tape stores where they were: bf16 parity, the bench's other_configs, rb parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04z6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rb.py tests/test_gpu_bf16.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
python bench.py > $O/r04_bench.json 2> $O/b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04z6/r04_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d.get("train_step_ms"), {k:v.get("ms_per_step") for k,v in d["other_configs"].items()})
PY
