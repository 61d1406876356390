# round 6, config C analysis: pack ranges A/B, kernel stats, SQ / traffic counters of the wavefront launches and the batched weight-gradient kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
ks() { local name=$1; shift; local out=$1; shift; local title=$1; shift
  ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o k -- "$@" ) > $O/$name.log 2>&1 < /dev/null
  python tools/rocprof_summary.py $(db $name) $O/$out "$title" > /dev/null 2>&1; rm -rf $O/$name; }
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
python -m pytest tests/test_gpu_stack.py tests/test_dropout.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
C="python3 $R/tools/probes/run_c.py"
ks c_slim c_slim.csv "config C, pack ranges (default)" $C
VMLMF_PACK_SLIM=0 ks c_full c_full.csv "config C, VMLMF_PACK_SLIM=0" $C
for w in 32 96 128 192 384; do VMLMF_WMIN=$w ks c_w$w c_w$w.csv "config C, VMLMF_WMIN=$w" $C; done
grep -h "wgrad_mfma_stack\|pack_stack" $O/c_*.csv | cut -c1-90
run_pmc u1 "$P1" $C; run_pmc u2 "$P2" $C; run_pmc uf "FETCH_SIZE" $C; run_pmc uw "WRITE_SIZE" $C
python tools/rocprof_pmc_util.py $O/pmc_util_config_c.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 tools/probes/run_c.py" $(db u1) $(db u2) > /dev/null 2>&1
python tools/rocprof_pmc.py $(db uf) $(db uw) $O/pmc_traffic_config_c.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/probes/run_c.py" > /dev/null 2>&1
rm -rf $O/u1 $O/u2 $O/uf $O/uw
bash tools/probes/r06_ab_c.sh > $O/ab_c.txt 2>&1; cat $O/ab_c.txt
