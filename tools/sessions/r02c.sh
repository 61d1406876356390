#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_rb.py -q -x ) > $O/pytest_rb.log 2>&1; echo "rc=$?" >> $O/pytest_rb.log
tail -40 $O/pytest_rb.log
