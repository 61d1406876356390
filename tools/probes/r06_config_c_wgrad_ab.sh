# round 6, config C: the four-tile weight-gradient kernel - parity tests, kernel stats with / without, A/B against round 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
timeout 900 python -m pytest tests/test_gpu_stack.py tests/test_dropout.py tests/test_gpu_modules.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -5 > $O/tests2.txt; cat $O/tests2.txt
C="python3 $R/tools/probes/run_c.py"
ks c4 c_wgrad4.csv "config C, wgrad4" $C
VMLMF_WGRAD4=0 ks c4off c_wgrad4_off.csv "config C, VMLMF_WGRAD4=0" $C
for w in 32 128; do VMLMF_WMIN=$w ks c4_w$w c_wgrad4_w$w.csv "config C, wgrad4, VMLMF_WMIN=$w" $C; done
grep -h "wgrad\|reduce_cg" $O/c_wgrad4*.csv | cut -c1-90
bash tools/probes/r06_ab_c.sh > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
