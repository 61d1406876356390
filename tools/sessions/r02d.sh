#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02d; mkdir -p $O
timeout 900 python tools/bench_rb.py e > $O/ab.jsonl 2> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
