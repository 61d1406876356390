#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02f; mkdir -p $O
for S in 16 8 4; do
  VMLMF_RB_S=$S VMLMF_LIB=$GRAFT_REPO_ROOT/tools/microbench/bin/libvmlmf_hip_stamp.so timeout 300 python tools/run_e.py --nograph 2>&1 | sort | uniq -c | sort -rn | head -4
done > $O/stamps.txt 2>&1
cat $O/stamps.txt | cut -c1-600
