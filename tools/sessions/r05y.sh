#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
timeout 1200 python -m pytest tests/test_gpu_rb.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do timeout 200 python tools/run_e.py --nograph 2>/dev/null | tail -1; done
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/kse -o k -- python3 $R/tools/run_e.py --nograph ) > $O/kse.log 2>&1 < /dev/null
python tools/rocprof_summary.py $(db kse) $O/r05_config_e_layer_kernel_stats.csv "config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), round 5 (final tree): rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph" > /dev/null 2>&1
rm -rf $O/kse
head -14 $O/r05_config_e_layer_kernel_stats.csv | cut -c1-110
