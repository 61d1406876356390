cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E |FAILED|passed|failed" | head
