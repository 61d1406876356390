#!/bin/bash
# r04z3: scope bits of the cluster exchange's tile stores / gather loads (rbvarF: both agent scope, G: loads, H: stores): parity + time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04z3; mkdir -p $O
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/base.so
for v in ${VARS:-base F G H}; do
  if [ $v = base ]; then cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so; else cp vmlmf_amd/lib/rbvar$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  timeout 600 python -m pytest tests/test_gpu_rb.py -x -q -m gpu > $O/tests_$v.txt 2>&1; echo "$v tests rc=$? $(grep -E 'passed|failed' $O/tests_$v.txt | tail -1)"
  for rep in 1 2; do
  echo "$v: group $(timeout 300 python tools/run_e.py 2>/dev/null | tail -1 | cut -c1-40) | v3 $(timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1 | cut -c1-40) | b32 $(timeout 300 python tools/run_e.py --batch 32 2>/dev/null | tail -1 | cut -c1-40)"
  done
done
cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so
