#!/bin/bash
# r04n: full GPU tier after wgrad_ring_kernel + layout change; refreshed config E / LM profiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04n4; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -30
[ $rc = 0 ] || exit 1
python bench.py --config E > $O/r04_config_e_1gpu.json 2> $O/e.err; tail -c 600 $O/r04_config_e_1gpu.json; echo
python bench.py > $O/r04_bench.json 2> $O/b.err; python - <<'EOF'
import json
d=json.loads(open('gpurun_out/r04n4/r04_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline'), {k:v for k,v in d.get('other_configs',{}).items()})
EOF
timeout 300 python tools/bench_lm.py > $O/r04_lm.jsonl 2>/dev/null; cut -c1-230 $O/r04_lm.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/pe -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $O/pe -name "*.db" | head -1) $O/r04_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), round 4 with wgrad_ring_kernel: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" 2>$O/sum.err | head -16 | cut -c1-150
rm -rf $O/pe
