cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02cm; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
prof() { tag=$1; shift
  ( cd /tmp && env "$@" BENCH_NOCPU=1 BENCH_ONLY="A-group" timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $R/$O/$tag -o k -- python3 $R/tools/bench_configs.py ) > $O/$tag.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $tag) $O/$tag.csv "$tag" > /dev/null 2>&1
  rm -rf $O/$tag
  echo "== $tag"; head -10 $O/$tag.csv | tail -7 | cut -c1-100
}
prof ride
prof off VMLMF_WRIDE=0
