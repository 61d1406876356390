#!/bin/bash
# r04f: the whole GPU tier after the round's changes so far, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gputests.txt | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" $O/gputests.txt | head -20
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04f/bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "ms_per_step_kept_images", "train_step_ms", "eager_ms_per_step")})
print(j["roofline"]["frac"], j["roofline"]["launch_us"], j["kernels_us"])
print(j.get("other_configs")); print(j.get("cpu_baseline", {}).get("value"))
PY
