#!/bin/bash
R=$GRAFT_REPO_ROOT; O=gpurun_out
cd $R; mkdir -p $O
export TMPDIR=/tmp
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
db() { find $O/$1 -name "*.db" | head -1; }
CMD="python3 $R/tools/run_c.py"
export VMLMF_STACK=1
run_pmc wf "FETCH_SIZE" $CMD; run_pmc ww "WRITE_SIZE" $CMD
python tools/rocprof_pmc.py $(db wf) $(db ww) $O/r02_pmc_traffic_config_c_wavefront.json "VMLMF_STACK=1 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 tools/run_c.py; config C, wavefront launches" > /dev/null 2>&1
rm -rf $O/wf $O/ww
cat $O/r02_pmc_traffic_config_c_wavefront.json
