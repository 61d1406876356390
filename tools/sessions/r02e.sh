#!/bin/bash
# r02e: A/B table UCI + C (row-block vs VALU), rocprof kernel stats of config E on the clustered row-block kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python tools/bench_rb.py uci > $O/ab_uci.jsonl 2> $O/ab.err
timeout 600 python tools/bench_rb.py c > $O/ab_c.jsonl 2>> $O/ab.err
timeout 600 python tools/bench_rb.py e32 > $O/ab_e32.jsonl 2>> $O/ab.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_e -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph ) > $O/prof_e.log 2>&1
python tools/rocprof_summary.py $(find $O/prof_e -name "*.db" | head -1) $O/r02_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), clustered row-block kernels: rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
rm -rf $O/prof_e
cat $O/ab_uci.jsonl $O/ab_c.jsonl $O/ab_e32.jsonl | cut -c1-330
