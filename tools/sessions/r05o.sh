#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05o; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/ks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db ks) $O/r05_kernel_stats_probe.csv "probe" > /dev/null 2>&1
rm -rf $O/ks
head -30 $O/r05_kernel_stats_probe.csv | cut -c1-160
tail -3 $O/ks.log
