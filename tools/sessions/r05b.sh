#!/bin/bash
# r05b: the pruned library (no ABL / STW7 / NO_HEAD_RIDE branches, no rb_xfold / rb_wgrad / two-rows rec4 / rec3 XM forms): GPU tier + headline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-extra > $O/b.json 2> $O/b.err
  python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/b.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), d.get("eager_ms_per_step"), {k:v for k,v in d["kernels_us"].items() if v})
PY
done
python bench.py --no-cpu-baseline > $O/b_full.json 2> $O/b_full.err; python -c "
import json; d=json.loads(open('gpurun_out/r05b/b_full.json').read().strip().splitlines()[-1]); print({k:(v.get('ms_per_step'), v.get('error')) for k,v in d['other_configs'].items()})"
