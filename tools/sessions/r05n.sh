#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05n
timeout 900 python bench.py > gpurun_out/r05n/bench.json 2> gpurun_out/r05n/bench.err
tail -3 gpurun_out/r05n/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05n/bench.json').read().strip().splitlines()[-1])
for k in ("value","ms_per_step","train_step_ms","eager_ms_per_step","harness","riding_workers","kernels_us","other_configs","roofline","cpu_baseline"):
    print(k, d.get(k))
PY
