#!/bin/bash
# r04l: wgrad_ring_kernel, second form (compile-time LDS strides, mask-free main loop): parity, kernel time, layer times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04l; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_wring.py -x -q -m gpu > $O/tests_wring.txt 2>&1; rc=$?; echo "wring tests rc=$rc"; grep -E "passed|failed" $O/tests_wring.txt | tail -2; grep -E "^(FAILED|ERROR)|^E  " $O/tests_wring.txt | head -30
[ $rc = 0 ] || exit 1
kstats() { python3 - "$1" <<'EOF'
import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + '/*.db')[0]
c = sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
for r in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc limit 8"): print("   %-70s %4d %8.1f us" % (r[0][:70], r[1], r[2]))
EOF
}
cd /tmp && export TMPDIR=/tmp
for v in "" "--v3"; do
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph $v > /dev/null 2>&1
echo "== run_e $v"; kstats $GRAFT_REPO_ROOT/$O/p; rm -rf $GRAFT_REPO_ROOT/$O/p
done
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in 0 -1; do
  echo "== VMLMF_WRING=$m"
  VMLMF_WRING=$m timeout 300 python tools/run_e.py 2>/dev/null | tail -1
  VMLMF_WRING=$m timeout 300 python tools/run_e.py --v3 2>/dev/null | tail -1
  VMLMF_WRING=$m timeout 300 python tools/bench_lm.py 2>/dev/null | head -1 | cut -c1-220
done
done
