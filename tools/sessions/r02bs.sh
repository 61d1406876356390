#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bs; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
prof() { tag=$1; shift
  ( cd /tmp && env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/$O/$tag -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra ) > $O/$tag.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $tag) $O/$tag.csv "$tag" > /dev/null 2>&1
  rm -rf $O/$tag
  echo "== $tag bwd $(grep rec_bwd $O/$tag.csv | cut -d, -f8-11) $(grep ms_per_step $O/$tag.log | grep -o '"ms_per_step": [0-9.]*')"
}
prof dry VMLMF_WRIDE_DRY=1
prof dry_end_wbl2 VMLMF_WRIDE_DRY=9
prof dry_wbl2_each2 VMLMF_WRIDE_DRY=41
prof dry_wbl2_each VMLMF_WRIDE_DRY=43
for e in "VMLMF_WRIDE_DRY=8" "VMLMF_WRIDE_DRY=40"; do echo "== $e"; env $e timeout 200 python tools/sessions/r02br.py 2>&1 | grep -A8 "^it 2" | cut -c1-100; done
