#!/bin/bash
# r04z5: rb_fwd_kernel with and without its tape stores (training vs torch.no_grad forward): are the exchange's vmcnt(0) waits paying for them?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04z5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "" "--infer" "--v3" "--v3 --infer"; do
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph $v > /dev/null 2>&1
echo "== run_e $v"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $GRAFT_REPO_ROOT/$O/p -name "*.db" | head -1) /tmp/x.csv "x" 2>/dev/null | grep -E "rb_fwd|rb_bwd|xexp" | cut -c1-110
rm -rf $GRAFT_REPO_ROOT/$O/p
done
