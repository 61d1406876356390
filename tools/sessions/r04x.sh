#!/bin/bash
# r04x: kernel split of a config E layer at 32 rows per GPU (the 8-GPU point of configs[4])
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04x2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "" "--v3"; do
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph --batch 32 $v > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $GRAFT_REPO_ROOT/$O/p -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/r04_config_e_layer_b32_kernel_stats$v.csv "config E layer at 32 rows per GPU $v (B=32, T=35, H=650): rocprofv3 --kernel-trace --stats -- python3 tools/run_e.py --nograph --batch 32 $v" 2>/dev/null | head -18 | cut -c1-130
rm -rf $GRAFT_REPO_ROOT/$O/p
done
