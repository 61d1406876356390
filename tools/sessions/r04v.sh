#!/bin/bash
# r04v: randomised parity with wgrad_ring_kernel forced wherever it takes the layer (big shapes: H up to 700, up to 1100 rows), and rb mode
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04v; mkdir -p $O
VMLMF_WRING=1 timeout 1200 python tools/fuzz_parity.py 80 41 big > $O/fuzz_big_wring1.txt 2>&1; echo "big rc=$?"; tail -3 $O/fuzz_big_wring1.txt | cut -c1-300
VMLMF_WRING=1 timeout 900 python tools/fuzz_parity.py 150 42 rb > $O/fuzz_rb_wring1.txt 2>&1; echo "rb rc=$?"; tail -3 $O/fuzz_rb_wring1.txt | cut -c1-300
timeout 900 python tools/fuzz_parity.py 60 43 big > $O/fuzz_big_auto.txt 2>&1; echo "big auto rc=$?"; tail -2 $O/fuzz_big_auto.txt | cut -c1-300
