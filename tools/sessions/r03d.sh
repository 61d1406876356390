#!/bin/bash
# r03d: riding workers with the faster backward rows: progress-word lag and rows per chunk
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03d; mkdir -p $O
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 200 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['eager_ms_per_step'], d['loss'])"; }
run VMLMF_REC3=2
for lag in 2 4 5 6 8; do run VMLMF_REC3=2 VMLMF_WRIDE_LAG=$lag; done
for rc in 16 64; do run VMLMF_REC3=2 VMLMF_WRIDE_RC=$rc; done
run VMLMF_REC3=2 VMLMF_WRIDE=0
run VMLMF_REC3=0 VMLMF_WRIDE=0
run VMLMF_REC3=2 VMLMF_WRIDE_DRY=2
run VMLMF_REC3=2 VMLMF_WRIDE_DRY=1
