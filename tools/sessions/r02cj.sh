cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wride.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -3
BENCH_NOCPU=1 BENCH_ONLY="A-group" timeout 300 python tools/bench_configs.py 2>/dev/null | grep config | cut -c1-160
