#!/bin/bash
# r03e: new parity tests (group cells on the wavefront vs oracle, demo shapes, config E at its own size), failure path
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wride.py tests/test_gpu_stack.py -x -q -m gpu -k "gives_up or against_the_fp64_oracle or demo_shapes" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_rb.py -x -q -m gpu -k "two_layers_at_its_own_size" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -s -m gpu -k "golden" 2>&1 | grep -E "config C bf16|passed|failed" | tail -30
