#!/bin/bash
# r04m: wgrad_ring_kernel ablation builds (vmlmf_amd/lib/ablK.so = the library with vmlmf_wgrad_ring.hip compiled -DWR_ABL=K): what bounds a stage?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04m; mkdir -p $O
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/abl0.so
cd /tmp && export TMPDIR=/tmp
for k in ${ABLS:-0 1 2 3 4 5}; do
  if [ $k = 0 ]; then cp /tmp/abl0.so $GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so; else cp $GRAFT_REPO_ROOT/vmlmf_amd/lib/abl$k.so $GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so; fi
  for v in "" "--v3"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/p -o e -- python3 $GRAFT_REPO_ROOT/tools/run_e.py --nograph $v > /dev/null 2>&1
  python3 - "$GRAFT_REPO_ROOT/$O/p" "$k$v" <<'EOF'
import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + '/*.db')[0]
c = sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
for r in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%wgrad_ring%' group by s.kernel_name"): print("abl", sys.argv[2], r[0][20:60], r[1], "%.1f us" % r[2])
EOF
  rm -rf $GRAFT_REPO_ROOT/$O/p
  done
done
cp /tmp/abl0.so $GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so
