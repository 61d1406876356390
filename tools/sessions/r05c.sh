#!/bin/bash
# r05c: same-box A/B: the tree of the end of round 4 (.abtree/old) against this one (pruned library + the criterion riding on the
# forward launch); then the new tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), d.get("eager_ms_per_step"), {k:v for k,v in d["kernels_us"].items() if v}, d["loss"])
PY
}
for rep in 1 2 3; do
  (cd .abtree/old && python bench.py --no-cpu-baseline --no-extra > $O/old.json 2> $O/old.err); show old $O/old.json
  python bench.py --no-cpu-baseline --no-extra --separate-loss > $O/sep.json 2> $O/sep.err; show new_separate_loss $O/sep.json
  python bench.py --no-cpu-baseline --no-extra > $O/new.json 2> $O/new.err; show new_fused_loss $O/new.json
done
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "criterion_riding or both_bindings or graphed or classifier_riding or net_adam" 2>&1 | tail -15
