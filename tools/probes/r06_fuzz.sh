# randomised parity on the final tree: sequence entry point, wavefront stacks (incl. layers of different sizes), row blocks, large layers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
{ timeout 900 python tools/fuzz_parity.py 400 61 seq 2>&1 | tail -4
  timeout 900 python tools/fuzz_parity.py 240 62 stack 2>&1 | tail -6
  timeout 900 python tools/fuzz_parity.py 200 63 rb 2>&1 | tail -4
  timeout 900 python tools/fuzz_parity.py 60 64 big 2>&1 | tail -4; } > $O/fuzz_parity.txt 2>&1
cat $O/fuzz_parity.txt
