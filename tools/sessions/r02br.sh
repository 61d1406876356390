cd $GRAFT_REPO_ROOT
for e in "VMLMF_WRIDE_DRY=1" "VMLMF_WRIDE_DRY=1 VMLMF_EXP_SYNC=1" "VMLMF_WRIDE_DRY=0 VMLMF_EXP_SYNC=1"; do echo "== $e"; env $e timeout 200 python tools/sessions/r02br.py 2>&1 | grep "^it" | cut -c1-230; done
