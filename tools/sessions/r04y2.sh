#!/bin/bash
# r04y2: cluster exchange with an epoch word per (member, wave) and no barrier between stores and publication
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04y2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rb.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests.txt | head -20
[ $rc = 0 ] || exit 1
for i in 1 2; do
timeout 300 python tools/run_e.py 2>/dev/null | tail -1
timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1
timeout 300 python tools/run_e.py --batch 32 2>/dev/null | tail -1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $GRAFT_REPO_ROOT/$O/p -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/e.csv "x" 2>/dev/null | grep -E "rb_fwd|rb_bwd" | cut -c1-120
rm -rf $GRAFT_REPO_ROOT/$O/p
