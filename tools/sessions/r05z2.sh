#!/bin/bash
# r05z2: the numbers that moved after r05z (rb_bwd_kernel's dropout hook, the dqx image, the embedding gather with dropout): bench line, LM steps, config E lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05z2; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py 256 --dropout 0.5 > $O/lm_dropout.jsonl 2>/dev/null < /dev/null
timeout 600 python bench.py --config E --steps 20 --warmup 5 > $O/config_e_lm_1gpu.json 2>/dev/null < /dev/null
for bp in 128 64; do timeout 600 python bench.py --config E --batch-per-gpu $bp --steps 20 --warmup 5 2>/dev/null < /dev/null; done > $O/config_e_lm_1gpu_b128_b64.jsonl
timeout 600 python bench.py --config E --batch-per-gpu 32 --steps 20 --warmup 5 > $O/config_e_lm_1gpu_b32.json 2>/dev/null < /dev/null
timeout 900 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksl -o k -- python3 $R/tools/bench_lm.py ) > $O/ksl.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksl) $O/r05_lm_step_kernel_stats.csv "tools/bench_lm.py: whole LM steps at config E's shape (13 steps each of: head in place V3 / group, two-call loss, stock, group), round 5 final tree: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/ksl
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05z2/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d.get("ms_per_step_kept_images"), d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r["achieved"], r["frac"], r["launch_us"], r.get("traffic"), d["cpu_baseline"]["value"], d.get("harness"))
print("other", {k:v["ms_per_step"] for k,v in d["other_configs"].items()})
for f in ("lm.jsonl","lm_dropout.jsonl","config_e_lm_1gpu.json","config_e_lm_1gpu_b128_b64.jsonl","config_e_lm_1gpu_b32.json","configs.jsonl"):
    for l in open("gpurun_out/r05z2/"+f):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, {k:j[k] for k in j if k in ("config","B","ms_per_step","ms_hipgraph","ms_eager","train_step_ms","head_in_place","fused_loss_and_update","ms_per_step_eager","dropout","dropout_launches","ms_p0","ms_package","ms_nn_dropout")} if "workload" not in str(j.get("config")) else (j["config"].get("batch_per_gpu"), j["ms_per_step"], j.get("train_step_ms")))
PY
