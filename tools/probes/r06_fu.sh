# round 6: the one finishing launch of a stack - parity, kernel stats with / without, config C A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
timeout 1500 python -m pytest tests/test_gpu_stack.py tests/test_gpu_rehearsal.py tests/test_dropout.py tests/test_gpu_modules.py tests/test_gpu_rbx.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -5 > $O/tests_fu.txt; cat $O/tests_fu.txt
C="python3 $R/tools/probes/run_c.py"
ks f1 c_fu.csv "config C, one finishing launch" $C
VMLMF_FINISH_UNITS=0 ks f0 c_fu_off.csv "config C, VMLMF_FINISH_UNITS=0" $C
grep -h "finish\|reduce_cg" $O/c_fu.csv $O/c_fu_off.csv | cut -c1-90
bash tools/probes/r06_ab_c.sh > $O/ab_c_fu.txt 2>&1; cat $O/ab_c_fu.txt
python tools/probes/rbx_probe.py 32 --plain --stacked-only; VMLMF_FINISH_UNITS=0 python tools/probes/rbx_probe.py 32 --plain --stacked-only
