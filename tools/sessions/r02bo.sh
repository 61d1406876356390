#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bo; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
prof() { tag=$1; shift
  ( cd /tmp && env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/$tag -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/$tag.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $tag) $O/$tag.csv "$tag" > /dev/null 2>&1
  rm -rf $O/$tag
  echo "== $tag"; head -9 $O/$tag.csv | tail -6 | cut -c1-110
}
export -f prof db
prof dry VMLMF_WRIDE_DRY=1
prof ride VMLMF_WRIDE_K=16
prof off VMLMF_WRIDE=0
