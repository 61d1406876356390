#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bv; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
prof() { tag=$1; shift
  ( cd /tmp && env "$@" timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $R/$O/$tag -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/$tag.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $tag) $O/$tag.csv "$tag" > /dev/null 2>&1
  rm -rf $O/$tag
  echo "== $tag bwd $(grep rec_bwd $O/$tag.csv | cut -d, -f8-11) $(grep ms_per_step $O/$tag.log | grep -o '"ms_per_step": [0-9.]*')"
}
timeout 60 python tools/sessions/r02br.py 2>&1 | grep "^it\|fault" | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head
prof ride
prof ride_k24 VMLMF_WRIDE_K=24
prof ride_k32 VMLMF_WRIDE_K=32
prof ride_k32_lag3 VMLMF_WRIDE_K=32 VMLMF_WRIDE_LAG=3
prof off VMLMF_WRIDE=0
