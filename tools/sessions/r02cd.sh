cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/vmlmf_amd/lib/var/lib_U32.so
VMLMF_LIB=$L timeout 60 python tools/sessions/r02br.py 2>&1 | grep "^it 2\|fault" | cut -c1-120
for rc in 32 64; do echo "U32 rc=$rc"; VMLMF_LIB=$L VMLMF_WRIDE_RC=$rc timeout 120 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
echo "U16 (shipping lib, ctypes)"; VMLMF_LIB=$GRAFT_REPO_ROOT/vmlmf_amd/lib/libvmlmf_hip.so timeout 120 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
