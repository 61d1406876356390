#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02g; mkdir -p $O
( time python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_cpp.json 2> $O/bench_cpp.err; python bench.py --no-cpu-baseline --repack > $O/bench_repack.json 2> $O/bench_repack.err
VMLMF_PYBIND=ctypes python bench.py --no-cpu-baseline > $O/bench_ctypes.json 2> $O/bench_ctypes.err
python - <<'PY'
import json
for f in ("bench_cpp","bench_repack","bench_ctypes"):
    try:
        d=json.loads(open(f"gpurun_out/r02g/{f}.json").read())
        print(f, "graph", d["ms_per_step"], "eager", d["eager_ms_per_step"], "train_step", d.get("train_step_ms"))
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r02g/{f}.err").read()[-1500:])
PY
