#!/bin/bash
# r02zz3: numbers after the weight-gradient workers moved into the recurrent backward launch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02zz3; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wride.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
BENCH_ONLY=UCI timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
db() { find $O/$1 -name "*.db" | head -1; }
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r02_zz3_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline (config A; eager region + hipGraph replays + untimed breakdown pass), weight-gradient workers riding on rec_bwd_kernel: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/ks
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02zz3/bench.json")); print("bench", d["value"], d["ms_per_step"], d["eager_ms_per_step"], d.get("train_step_ms"), d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("speedup_vs_cpu"))
for f in ("configs.jsonl",):
    for l in open("gpurun_out/r02zz3/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","B","ms_hipgraph","ms_eager")})
PY
head -12 $O/r02_zz3_kernel_stats.csv | cut -c1-110
