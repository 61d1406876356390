cd $GRAFT_REPO_ROOT
BENCH_NOCPU=1 BENCH_ONLY="A-group" timeout 300 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-160
BENCH_NOCPU=1 BENCH_ONLY="A-group" timeout 300 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-160
timeout 200 python bench.py --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
timeout 900 python -m pytest tests/test_gpu_wride.py -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -3
