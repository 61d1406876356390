#!/bin/bash
# r04z2: polling variants of the cluster exchange (vmlmf_amd/lib/rbvarX.so): B = wave 0 alone polls (+ barrier), C = no s_sleep, D = s_sleep 4
cd "$GRAFT_REPO_ROOT" || exit 1
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/base.so
for rep in 1 2; do for v in base B C D; do
  if [ $v = base ]; then cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so; else cp vmlmf_amd/lib/rbvar$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  echo "$v: group $(timeout 300 python tools/run_e.py 2>/dev/null | tail -1 | cut -c1-40) | v3 $(timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1 | cut -c1-40) | b32 $(timeout 300 python tools/run_e.py --batch 32 2>/dev/null | tail -1 | cut -c1-40)"
done; done
cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so
