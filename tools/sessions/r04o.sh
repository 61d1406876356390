#!/bin/bash
# r04o: wgrad_ring_kernel below 2048 rows (config E at 32 / 64 / 128 rows per GPU): where does the automatic choice belong?
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 32 64 128; do for m in -1 1; do
  echo "== B=$b VMLMF_WRING=$m  $(VMLMF_WRING=$m timeout 300 python tools/run_e.py --batch $b 2>/dev/null | tail -1)  | v3: $(VMLMF_WRING=$m timeout 300 python tools/run_e.py --v3 --batch $b 2>/dev/null | tail -1)"
done; done
