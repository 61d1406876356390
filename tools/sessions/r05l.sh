#!/bin/bash
# round 5, session l: dropout without mask tensors (ABI 11) - parity tests, the LM step at p = 0.5
cd /root/repo
mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_dropout.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r05l/tests.txt
cat gpurun_out/r05l/tests.txt
timeout 300 python tools/bench_lm.py 256 --dropout 0.5 2>&1 | tail -6 | tee gpurun_out/r05l/bench_lm.txt
timeout 300 python tools/bench_lm.py 32 --dropout 0.5 2>&1 | tail -2 | tee -a gpurun_out/r05l/bench_lm.txt
