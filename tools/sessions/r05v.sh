#!/bin/bash
# randomised parity on the final tree of round 5: sequence entry point (direct mode / finish2 / riding criterion are the default forms),
# wavefront stacks, row-block kernels forced, large shapes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05v; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py 500 501 seq 2>&1 | tail -4 > $O/seq.txt; cat $O/seq.txt
timeout 900 python tools/fuzz_parity.py 120 502 stack 2>&1 | tail -4 > $O/stack.txt; cat $O/stack.txt
timeout 1200 python tools/fuzz_parity.py 300 503 rb 2>&1 | tail -4 > $O/rb.txt; cat $O/rb.txt
timeout 1800 python tools/fuzz_parity.py 100 504 big 2>&1 | tail -4 > $O/big.txt; cat $O/big.txt
