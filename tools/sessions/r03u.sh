#!/bin/bash
# r03u: hipGraph capture in thread-local mode with a live RCCL process group (one rank, --force-collective), eager all-reduce after
# each replay and (second run) captured inside the graph
cd "$GRAFT_REPO_ROOT" || exit 1
for extra in "" "--graph-collective"; do
  timeout 500 python bench.py --force-collective --transport torch $extra --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2> gpurun_out/fc.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$extra', j['ms_per_step'], j['config']['launch'], j['config']['allreduce_transport'], j['config']['rccl_ranks'], j['config']['collectives_per_step'], j.get('allreduce_ms'))
"
  grep -i "capture\|error\|fail" gpurun_out/fc.err | head -5
done
