#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 60 python tools/sessions/r02br.py 2>&1 | grep "^it 2\|fault" | cut -c1-120
timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-200
BENCH_NOCPU=1 BENCH_ONLY=UCI timeout 900 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-160
