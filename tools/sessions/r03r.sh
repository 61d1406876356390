#!/bin/bash
# r03r: rehearsal of the N > 1 path of bench.py on the one-GPU box: 2 and 4 ranks sharing GPU 0, gradients over gloo
# (VMLMF_BENCH_REHEARSAL=1), started both ways the driver may start it (plain --gpus N, and torch.distributed.run)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03r
export VMLMF_BENCH_REHEARSAL=1 VMLMF_BENCH_RANK_TIMEOUT=500
echo "== plain --gpus 2"
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03r/plain2.json 2> gpurun_out/r03r/plain2.err; echo "rc=$?"
tail -c 1500 gpurun_out/r03r/plain2.json; tail -5 gpurun_out/r03r/plain2.err
echo "== torchrun 4 ranks (VMLMF_WRIDE=0: four processes on one GPU starve the riding weight-gradient workers of their rows - see DESIGN.md section 7)"
VMLMF_WRIDE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03r/run4.json 2> gpurun_out/r03r/run4.err; echo "rc=$?"
tail -c 1500 gpurun_out/r03r/run4.json; tail -5 gpurun_out/r03r/run4.err
echo "== strong, 2 ranks"
timeout 600 python bench.py --gpus 2 --global-batch 512 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03r/strong2.json 2> gpurun_out/r03r/strong2.err; echo "rc=$?"
tail -c 800 gpurun_out/r03r/strong2.json; tail -5 gpurun_out/r03r/strong2.err
