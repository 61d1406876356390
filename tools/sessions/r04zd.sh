#!/bin/bash
# r04zd: two PTB layers, time chunks of the layers on two streams (tools/experiments/chunk_pipeline_probe.py)
cd "$GRAFT_REPO_ROOT" || exit 1
for a in "" "--chunks 3" "--chunks 5" "--v3" "--v3 --chunks 3"; do echo "== $a"; timeout 300 python tools/experiments/chunk_pipeline_probe.py $a 2>&1 | grep -v "amdgpu.ids" | tail -2; done
