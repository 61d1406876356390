#!/bin/bash
# r04u: sgd_clip_kernel with 16-byte accesses / no gradient write-back when unclipped, sqsum with 1024-thread workgroups
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
[ $rc = 0 ] || exit 1
for i in 1 2 3; do timeout 300 python tools/bench_lm.py --only-head 2>/dev/null | head -1 | cut -c1-220; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/bench_lm.py --only-head > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $O/p -name "*.db" | head -1) $O/r04_lm_step_kernel_stats.csv "tools/bench_lm.py --only-head: 13 LM steps (Model.loss + clip_sgd_step, 2 x MyVMLSTM rank 32, B 256, T 35, vocabulary 10000) + the GEMM form timing of the first call, round 4 final: rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
grep -E "sqsum|sgd_clip|norm_kernel|transpose|nll_" $O/r04_lm_step_kernel_stats.csv | cut -c1-150
rm -rf $O/p
