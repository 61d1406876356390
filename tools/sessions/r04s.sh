#!/bin/bash
# r04s: same-box A/B of the headline: the library before wgrad_ring_kernel (commit 15b24f9) against the tree's
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04s; mkdir -p $O
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/new.so
for rep in 1 2; do for w in old new; do
  if [ $w = old ]; then cp vmlmf_amd/lib/old_15b24f9.so vmlmf_amd/lib/libvmlmf_hip.so; else cp /tmp/new.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  python bench.py --no-cpu-baseline --no-extra > $O/b.json 2> $O/b.err
  python - "$w" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/r04s/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("train_step_ms"), {k:v for k,v in d["kernels_us"].items() if v})
PY
done; done
cp /tmp/new.so vmlmf_amd/lib/libvmlmf_hip.so
