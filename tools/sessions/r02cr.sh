cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -3; done
timeout 300 python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu'])"
