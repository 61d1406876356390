cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2; do timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -3; done
