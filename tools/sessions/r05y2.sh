#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ch in 12 6 4 46 412; do for i in 1 2; do echo -n "ch=$ch "; VMLMF_DQX_CH=$ch timeout 200 python tools/run_e.py --nograph 2>/dev/null | tail -1; done; done
