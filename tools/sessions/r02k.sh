#!/bin/bash
# r02k: PMC passes (utilisation + HBM traffic) on the headline bench and on the config E layer; kernel stats of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
run_pmc() {  # name, counters..., then the program
  local name=$1; shift; local ctr=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d $R/$O/$name -o p -- "$@" ) > $O/$name.log 2>&1
}
BENCH="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph"
EL="python3 $R/tools/run_e.py --nograph"
run_pmc a1 "$P1" $BENCH; run_pmc a2 "$P2" $BENCH
run_pmc af "FETCH_SIZE" $BENCH; run_pmc aw "WRITE_SIZE" $BENCH
run_pmc e1 "$P1" $EL; run_pmc e2 "$P2" $EL
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc_util.py $O/r02_pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph; config A (B=64 T=128 H=180 r=16)" $(db a1) $(db a2) > $O/util_a.log 2>&1
python tools/rocprof_pmc_util.py $O/r02_pmc_util_config_e.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 tools/run_e.py --nograph; config E layer (V4 group, H=650, ranks 32/[32,32], B=256, T=35), clusters of 16" $(db e1) $(db e2) > $O/util_e.log 2>&1
python tools/rocprof_pmc.py $(db af) $(db aw) $O/r02_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph; config A; merged by tools/rocprof_pmc.py" > $O/traffic.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline ) > $O/ks.log 2>&1
python tools/rocprof_summary.py $(db ks) $O/r02_kernel_stats.csv "bench.py --steps 50 --warmup 10 --no-cpu-baseline (config A; eager region + hipGraph replays + untimed breakdown pass): rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
rm -rf $O/a1 $O/a2 $O/af $O/aw $O/e1 $O/e2 $O/ks
ls -la $O; head -12 $O/r02_kernel_stats.csv
