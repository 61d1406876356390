#!/bin/bash
# r05e: direct mode (register images from the reference layouts inside the recurrent kernels): parity, then same-box A/B against
# VMLMF_DIRECT=0 and the old tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inrow.py tests/test_gpu_wride.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q 2>&1 | tail -5
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), {k:v for k,v in d["kernels_us"].items() if v}, d["loss"])
PY
}
for rep in 1 2 3; do
  (cd .abtree/old && python bench.py --no-cpu-baseline --no-extra > $O/old.json 2> $O/old.err); show old $O/old.json
  VMLMF_DIRECT=0 python bench.py --no-cpu-baseline --no-extra > $O/nd.json 2> $O/nd.err; show direct0 $O/nd.json
  python bench.py --no-cpu-baseline --no-extra > $O/new.json 2> $O/new.err; show direct1 $O/new.json
done
