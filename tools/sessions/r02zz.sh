#!/bin/bash
# r02zz: numbers at the end of round 2 (after the wavefront launches and the MFMA x-side expansion of large layers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02zz; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
timeout 300 python tools/bench_rb.py e32 > $O/e32.jsonl 2>/dev/null < /dev/null
db() { find $O/$1 -name "*.db" | head -1; }
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r02_zz_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline (config A; eager region + hipGraph replays + untimed breakdown pass): rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ke -o k -- python3 $R/tools/run_e.py --nograph ) > $O/ke.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ke) $O/r02_zz_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), clusters of 16 on the row-block kernels, x side expanded on MFMA: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
rm -rf $O/ks $O/ke
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02zz/bench.json")); print("bench", d["value"], d["ms_per_step"], d["eager_ms_per_step"], d["train_step_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["speedup_vs_cpu"])
for f in ("configs.jsonl","lm.jsonl","e32.jsonl"):
    for l in open("gpurun_out/r02zz/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","shape","B","ms_per_step","ms_hipgraph","ms_eager","ms_fwd_bwd","ms_per_step_eager","fused_loss_and_update","speedup_vs_cpu")})
PY
head -12 $O/r02_zz_kernel_stats.csv | cut -c1-110
