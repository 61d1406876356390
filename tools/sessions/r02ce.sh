#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 60 python tools/sessions/r02br.py 2>&1 | grep "^it 2\|fault" | cut -c1-120
timeout 120 python tools/sessions/r02cc.py 2>&1 | grep -v "^  File\|^Extension\|amdgpu.ids" | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-200
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -5
BENCH_NOCPU=1 timeout 900 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-160
