#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_bf16.py -q -s ) > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
grep -E "bf16 \(|config C bf16|passed|failed|Error|err " $O/pytest_bf16.log | cut -c1-260 | tail -50
