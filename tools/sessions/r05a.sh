#!/bin/bash
# r05a: baseline of the round on a fresh box: headline as is, with rec3_fwd_kernel forced (VMLMF_REC3=7), GPU test tier
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a; mkdir -p $O
for w in default rec3fwd; do
  if [ $w = rec3fwd ]; then export VMLMF_REC3=7; else unset VMLMF_REC3; fi
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-extra > $O/b_$w.json 2> $O/b_$w.err
  python - "$w" "$O/b_$w.json" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), d.get("eager_ms_per_step"), {k:v for k,v in d["kernels_us"].items() if v})
PY
  done
done
unset VMLMF_REC3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
