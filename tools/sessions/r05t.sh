#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
timeout 900 python -m pytest tests/test_dropout.py -m gpu -q 2>&1 | tail -4
timeout 600 python tools/bench_lm.py 256 --dropout 0.5 2>/dev/null | tee $O/lm_dropout.jsonl | tail -1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ksd -o k -- python3 $R/tools/bench_lm.py 256 --dropout 0.5 ) > $O/ksd.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ksd) $O/r05_lm_step_dropout_kernel_stats.csv "tools/bench_lm.py 256 --dropout 0.5: 13 LM steps each at p = 0, at p = 0.5 with the package's mask-free dropout, at p = 0.5 with nn.Dropout (Model.stock_dropout: the 39 fused_dropout launches and their 39 masked-scale backward launches are that third phase's), group layers, round 5: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/ksd
grep -i "drop\|embed\|rb_fwd\|rb_bwd\|masked" $O/r05_lm_step_dropout_kernel_stats.csv | cut -c1-140
