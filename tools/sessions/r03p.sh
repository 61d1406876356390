#!/bin/bash
# r03p: criterion riding on the forward launch (vmlmf_head.target, Net.loss): parity tests, then bench with and without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03p
timeout 1500 python -m pytest tests/test_gpu_criterion.py -x -q -m gpu 2>&1 | tail -25
timeout 600 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r03p/bench_riding.json 2> gpurun_out/r03p/bench_riding.err
timeout 600 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --separate-loss > gpurun_out/r03p/bench_separate.json 2> gpurun_out/r03p/bench_separate.err
for f in riding separate; do python - <<PY
import json
l=[x for x in open("gpurun_out/r03p/bench_$f.json") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("$f", j["ms_per_step"], j.get("ms_per_step_kept_images"), j.get("train_step_ms"), j.get("eager_ms_per_step"), j["kernels_us"])
else:
    print("$f: no line"); print(open("gpurun_out/r03p/bench_$f.err").read()[-1500:])
PY
done
