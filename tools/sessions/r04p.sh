#!/bin/bash
# r04p: per-kernel split of the shipped LM step (Model.loss + clip_sgd_step, plain layers) after wgrad_ring_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/bench_lm.py --only-head > $GRAFT_REPO_ROOT/$O/line.txt 2>&1
cd $GRAFT_REPO_ROOT; cat $O/line.txt | tail -1 | cut -c1-200
python tools/rocprof_summary.py $(find $O/p -name "*.db" | head -1) $O/r04_lm_step_kernel_stats.csv "tools/bench_lm.py --only-head: 13 LM steps (Model.loss + clip_sgd_step, 2 x MyVMLSTM rank 32, B 256, T 35, vocabulary 10000), round 4 with wgrad_ring_kernel: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -45 $O/r04_lm_step_kernel_stats.csv | cut -c1-140
rm -rf $O/p
