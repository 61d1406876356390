# round 6, config C: counters of the four-tile weight-gradient kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
R=$GRAFT_REPO_ROOT
db() { find $O/$1 -name "*.db" | head -1; }
run_pmc() { local name=$1; shift; local ctr=$1; shift; ( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1 < /dev/null; }
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCP_PENDING_STALL_CYCLES_sum"
P5="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum"
C="python3 $R/tools/probes/run_c.py"
run_pmc v1 "$P1" $C; run_pmc v2 "$P2" $C; run_pmc v3 "$P3" $C; run_pmc v4 "$P4" $C; run_pmc v5 "$P5" $C; run_pmc vf "FETCH_SIZE" $C; run_pmc vw "WRITE_SIZE" $C
python tools/rocprof_pmc_util.py $O/pmc_util_config_c_wgrad4.json "wgrad4" $(db v1) $(db v2) $(db v3) $(db v4) $(db v5) > $O/util.log 2>&1
python tools/rocprof_pmc.py $(db vf) $(db vw) $O/pmc_traffic_config_c_wgrad4.json "wgrad4" > /dev/null 2>&1
tail -3 $O/v3.log $O/v4.log $O/v5.log
rm -rf $O/v1 $O/v2 $O/v3 $O/v4 $O/v5 $O/vf $O/vw
