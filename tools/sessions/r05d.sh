#!/bin/bash
# r05d: same-box A/B of loop alignment (-falign-loops=64 / 128 on the recurrent kernels' translation units) and of the criterion's
# prefetched targets; old tree as the reference
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), {k:v for k,v in d["kernels_us"].items() if v})
PY
}
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/base.so
for rep in 1 2 3; do
  (cd .abtree/old && python bench.py --no-cpu-baseline --no-extra > $O/old.json 2> $O/old.err); show old $O/old.json
  for v in base al64 al128; do
    if [ $v = base ]; then cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so; else cp vmlmf_amd/lib/alt_$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
    python bench.py --no-cpu-baseline --no-extra > $O/$v.json 2> $O/$v.err; show $v $O/$v.json
  done
done
cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so
