#!/bin/bash
# r04r: after moving the reduction's block counts out of VGeo: ring / rb parity, the headline twice
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wring.py tests/test_gpu_rb.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
[ $rc = 0 ] || exit 1
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04r/bench$i.json").read().strip().splitlines()[-1])
print("bench$i", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), d["kernels_us"], {k:v.get("ms_per_step") for k,v in d["other_configs"].items()})
PY
done
