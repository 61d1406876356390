#!/bin/bash
# r04z4: does the freshness of the gathered tiles cost? (rbvarS.so: -DRB_STALE gathers the tiles of the step before; wrong results, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/base.so
for v in base S base S; do
  if [ $v = base ]; then cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so; else cp vmlmf_amd/lib/rbvar$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  echo "$v: group $(timeout 300 python tools/run_e.py 2>/dev/null | tail -1 | cut -c1-40) | v3 $(timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1 | cut -c1-40)"
done
cp /tmp/base.so vmlmf_amd/lib/libvmlmf_hip.so
