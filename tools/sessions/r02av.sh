#!/bin/bash
# PMC passes at config C: wavefront launches (VMLMF_STACK=1) and chained per-layer kernels (VMLMF_STACK=0)
R=$GRAFT_REPO_ROOT; O=gpurun_out
cd $R; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
db() { find $O/$1 -name "*.db" | head -1; }
CMD="python3 $R/tools/run_c.py"
export VMLMF_STACK=1
run_pmc w1 "$P1" $CMD; run_pmc w2 "$P2" $CMD; run_pmc wf "FETCH_SIZE" $CMD; run_pmc ww "WRITE_SIZE" $CMD
python tools/rocprof_pmc_util.py $O/r02_pmc_util_config_c_wavefront.json "VMLMF_STACK=1 rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 tools/run_c.py; config C (2 x 256, r 24, B 128, T 24), wavefront launches" $(db w1) $(db w2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db wf) $(db ww) $O/r02_pmc_traffic_config_c_wavefront.json "VMLMF_STACK=1 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 tools/run_c.py; config C, wavefront launches" > /dev/null 2>&1
export VMLMF_STACK=0
run_pmc cf "FETCH_SIZE" $CMD; run_pmc cw "WRITE_SIZE" $CMD
python tools/rocprof_pmc.py $(db cf) $(db cw) $O/r02_pmc_traffic_config_c_chained.json "VMLMF_STACK=0 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 tools/run_c.py; config C, chained per-layer kernels" > /dev/null 2>&1
rm -rf $O/w1 $O/w2 $O/wf $O/ww $O/cf $O/cw
ls -la $O/r02_pmc_*config_c*
