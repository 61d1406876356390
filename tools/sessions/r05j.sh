#!/bin/bash
# r05j: where the other workloads stand after direct mode / finish2 / the riding criterion: strong-scaling points on one GPU, other_configs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
for gb in 128 256 512; do
  python bench.py --no-cpu-baseline --no-extra --global-batch $gb > $O/b$gb.json 2> $O/b$gb.err
  python - "$gb" "$O/b$gb.json" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("B", sys.argv[1], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), {k:v for k,v in d["kernels_us"].items() if v})
PY
done
python bench.py --no-cpu-baseline > $O/full.json 2> $O/full.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05j/full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["train_step_ms"], d["eager_ms_per_step"])
print({k:(v.get("ms_per_step"), v.get("chained_ms_per_step"), v.get("error")) for k,v in d["other_configs"].items()})
PY
