cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ch; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wride.py -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" | head -5
for gb in 512 256 128; do timeout 200 python bench.py --gpus 1 --global-batch $gb --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null < /dev/null; done > $O/bench_strong_1gpu.jsonl
python -c "
import json
for l in open('$O/bench_strong_1gpu.jsonl'):
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['config']['batch_per_gpu'], j['ms_per_step'], j.get('train_step_ms'))
"
