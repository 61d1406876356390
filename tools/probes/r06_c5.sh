# pack ranges A/B again (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
C="python3 $R/tools/probes/run_c.py"
for i in 1 2; do
ks p1 p_slim$i.csv "slim" $C; VMLMF_PACK_SLIM=0 ks p0 p_full$i.csv "full" $C
echo "slim: $(grep -h pack_stack $O/p_slim$i.csv | cut -c1-50)   full: $(grep -h pack_stack $O/p_full$i.csv | cut -c1-50)"
done
