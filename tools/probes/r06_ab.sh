# same-box A/B of the headline line: round-5 tree (gpurun_tmp/r05) against this tree, interleaved
for i in 1 2; do
  (cd gpurun_tmp/r05 && python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r05', d['ms_per_step'], d['train_step_ms'], d['eager_ms_per_step'], d['harness'])")
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r06', d['ms_per_step'], d['train_step_ms'], d['eager_ms_per_step'], d['harness'])"
done
