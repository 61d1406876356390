#!/bin/bash
# HBM traffic of config C in bf16 (row-block MFMA kernels): FETCH_SIZE / WRITE_SIZE passes
R=$GRAFT_REPO_ROOT; O=gpurun_out
cd $R; mkdir -p $O
export TMPDIR=/tmp
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
db() { find $O/$1 -name "*.db" | head -1; }
CMD="python3 $R/tools/run_c.py --bf16"
run_pmc bf "FETCH_SIZE" $CMD; run_pmc bw "WRITE_SIZE" $CMD
python tools/rocprof_pmc.py $(db bf) $(db bw) $O/r02_pmc_traffic_config_c_bf16.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 tools/run_c.py --bf16; config C, dtype bf16 (row-block MFMA kernels)" > /dev/null 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/bks -o k -- $CMD ) > $O/bks.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db bks) $O/r02_config_c_bf16_kernel_stats.csv "tools/run_c.py --bf16: config C, dtype bf16 (row-block MFMA kernels): rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/bf $O/bw $O/bks
cat $O/r02_pmc_traffic_config_c_bf16.json | grep -E "kernel\"|hbm"
head -12 $O/r02_config_c_bf16_kernel_stats.csv | cut -c1-100
