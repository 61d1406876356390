#!/bin/bash
# r05g: finish2_kernel (reduce + finish in one launch behind a riding backward): tests, then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wride.py tests/test_gpu_modules.py tests/test_gpu_inrow.py -x -q 2>&1 | grep -E "^E |FAILED|passed|failed" | head
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), {k:v for k,v in d["kernels_us"].items() if v}, d["loss"])
PY
}
for rep in 1 2 3; do
  (cd .abtree/old && python bench.py --no-cpu-baseline --no-extra > $O/old.json 2> $O/old.err); show old $O/old.json
  VMLMF_FINISH2=0 python bench.py --no-cpu-baseline --no-extra > $O/f0.json 2> $O/f0.err; show finish2_off $O/f0.json
  python bench.py --no-cpu-baseline --no-extra > $O/new.json 2> $O/new.err; show new $O/new.json
done
