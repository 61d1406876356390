#!/bin/bash
# r03f: bench with the new line (repack default, kept-image figure, rccl_ranks), forced single-rank collective, full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 python bench.py --force-collective --no-cpu-baseline --no-extra > $O/bench_forced.json 2> $O/bench_forced.err < /dev/null
timeout 300 python bench.py --keep-images --no-cpu-baseline --no-extra > $O/bench_keep.json 2>/dev/null < /dev/null
python - <<'PY'
import json
for f in ("bench","bench_forced","bench_keep"):
    try:
        d=json.load(open(f"gpurun_out/r03f/{f}.json"))
    except Exception as e:
        print(f, "FAILED", e); continue
    r=d["roofline"]; c=d["config"]
    print(f, d["ms_per_step"], d.get("ms_per_step_kept_images"), d.get("ms_per_step_repack"), d["train_step_ms"], d["eager_ms_per_step"], c.get("rccl_ranks"), c.get("collectives_per_step"), c.get("allreduce_transport"), d.get("allreduce_ms"), r["kernel"], r["frac"], r["recurrence_only_frac"], r["launch_workgroups_one_per_cu"], d["loss"])
    if "cpu_baseline" in d: print(d["cpu_baseline"]["cpu_model"], d["cpu_baseline"]["value"], d["speedup_vs_cpu"], d.get("other_configs"))
PY
tail -3 $O/bench.err $O/bench_forced.err
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
