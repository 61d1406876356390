#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bp; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
prof() { tag=$1; shift
  ( cd /tmp && env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/$tag -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/$tag.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $tag) $O/$tag.csv "$tag" > /dev/null 2>&1
  rm -rf $O/$tag
  echo "== $tag $(grep rec_bwd $O/$tag.csv | cut -d, -f8-11)"
}
prof dry_sys VMLMF_WRIDE_DRY=1
prof dry_sc1 VMLMF_WRIDE_DRY=3
prof dry_plain VMLMF_WRIDE_DRY=5
prof dry_nt VMLMF_WRIDE_DRY=7
prof dry_sys_fl4 VMLMF_WRIDE_DRY=9
prof dry_plain_fl4 VMLMF_WRIDE_DRY=13
prof dry_sys_lag8 VMLMF_WRIDE_DRY=1 VMLMF_WRIDE_LAG=8
