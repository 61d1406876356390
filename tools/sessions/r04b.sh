#!/bin/bash
# r04b: the new -m gpu tests (rehearsal of the N > 1 path incl. config E, optimizer gate, graph re-capture), then the whole GPU tier
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rehearsal.py tests/test_gpu_wride.py -x -q -m gpu > $O/new_tests.txt 2>&1; echo "new tests rc=$?"; tail -40 $O/new_tests.txt
VMLMF_BENCH_REHEARSAL=1 timeout 600 python bench.py --config E --gpus 2 --steps 10 --warmup 3 > $O/rehearsal_e_strong2.json 2> $O/rehearsal_e.err; echo "E rehearsal rc=$?"; cat $O/rehearsal_e_strong2.json; tail -3 $O/rehearsal_e.err
timeout 600 python bench.py --config E --steps 20 --warmup 5 > $O/e_1gpu.json 2> $O/e_1gpu.err; echo "E 1gpu rc=$?"; cat $O/e_1gpu.json; tail -3 $O/e_1gpu.err
timeout 600 python bench.py --config E --batch-per-gpu 32 --steps 20 --warmup 5 > $O/e_1gpu_b32.json 2>> $O/e_1gpu.err; cat $O/e_1gpu_b32.json
