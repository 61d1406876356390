#!/bin/bash
# r03ya: row-block / cluster forward (config E): gate and cell tapes stored with the non-temporal hint (libA) against shipped (libC)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in A C; do
  echo "$rep lib$v $(VMLMF_LIB=$GRAFT_REPO_ROOT/gpurun_in/lib$v.so timeout 300 python tools/run_e.py 2>&1 | tail -1) | $(VMLMF_LIB=$GRAFT_REPO_ROOT/gpurun_in/lib$v.so timeout 300 python tools/run_e.py --v3 2>&1 | tail -1)"
done
done
