# staged: a short parity run first (a faulting kernel must not sit in a core dump for minutes), then the tests and the timing
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 150 python -m pytest tests/test_gpu_stack.py -x -q -m gpu -k "matches_chained or different_hidden" 2>&1 | tail -3 > gpurun_out/c6_quick.txt; cat gpurun_out/c6_quick.txt
grep -q " passed" gpurun_out/c6_quick.txt || exit 1
grep -q "failed" gpurun_out/c6_quick.txt && exit 1
bash tools/probes/r06_c2.sh
