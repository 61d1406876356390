#!/bin/bash
# r03c: first run of the rec3 kernels in the library: bench (rec3 on / off), kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2> $O/bench.err < /dev/null
VMLMF_REC3=0 timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/bench_rec3off.json 2>/dev/null < /dev/null
VMLMF_REC3=2 timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/bench_rec3bwd.json 2>/dev/null < /dev/null
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(find $O/ks -name "*.db" | head -1) $O/kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra (config A), rec3 kernels: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/ks
for f in bench bench_rec3off bench_rec3bwd; do python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1], d["ms_per_step"], d["eager_ms_per_step"], d.get("train_step_ms"), r["kernel"], r.get("launch_us"), d.get("loss"))
PY
done
head -14 $O/kernel_stats.csv | cut -c1-150
