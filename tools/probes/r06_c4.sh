# round 6, config C: what the four-tile weight-gradient kernel spends its time on (compile-time ablations; results wrong, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
C="python3 $R/tools/probes/run_c.py"
for d in 0 1 2 3 4; do for w in 64; do VMLMF_WMIN=$w VMLMF_W4ABL=$d ks d$d c_w4abl${d}_$w.csv "config C, VMLMF_W4ABL=$d VMLMF_WMIN=$w" $C; echo "abl $d wmin $w: $(grep -h wgrad4 $O/c_w4abl${d}_$w.csv | cut -c1-60)"; done; done
