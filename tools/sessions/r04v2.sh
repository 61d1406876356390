#!/bin/bash
# r04v2: randomised parity on the final tree: sequences, stacks, row-block
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04v2; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 400 51 seq > $O/fuzz_seq.txt 2>&1; echo "seq rc=$?"; tail -1 $O/fuzz_seq.txt | cut -c1-300
timeout 900 python tools/fuzz_parity.py 120 52 stack > $O/fuzz_stack.txt 2>&1; echo "stack rc=$?"; tail -1 $O/fuzz_stack.txt | cut -c1-300
timeout 600 python tools/fuzz_parity.py 150 53 rb > $O/fuzz_rb.txt 2>&1; echo "rb rc=$?"; tail -1 $O/fuzz_rb.txt | cut -c1-300
