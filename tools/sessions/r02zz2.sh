#!/bin/bash
# r02zz2: large-layer numbers after the exchange changes (sparse tiles, 16-way gather)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02zz2; mkdir -p $O
R=$GRAFT_REPO_ROOT
BENCH_ONLY=PTB timeout 600 python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null < /dev/null
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2>/dev/null < /dev/null
timeout 300 python tools/bench_rb.py e32 > $O/e32.jsonl 2>/dev/null < /dev/null
db() { find $O/$1 -name "*.db" | head -1; }
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ke -o k -- python3 $R/tools/run_e.py --nograph ) > $O/ke.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ke) $O/r02_zz2_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), clusters of 16 on the row-block kernels (sparse exchange tiles, 16-way gather), x side on MFMA: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
rm -rf $O/ke
grep -h "config\|shape" $O/configs.jsonl $O/lm.jsonl $O/e32.jsonl | cut -c1-200
head -10 $O/r02_zz2_config_e_layer_kernel_stats.csv | cut -c1-100
