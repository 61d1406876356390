cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do timeout 900 python -m pytest tests/test_gpu_wride.py tests/test_gpu_stack.py -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -3; done
