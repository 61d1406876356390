cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-extra 2>/dev/null < /dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
timeout 900 python -m pytest tests/test_gpu_wride.py -x -q 2>&1 | grep -E "passed|failed|FAILED|rror|assert" | head -8
