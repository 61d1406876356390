#!/bin/bash
# GPU session r02b: full GPU test suite + PMC utilisation passes (program directly after --, absolute path)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
( cd /tmp && rocprofv3 --kernel-trace --pmc $P1 -d $GRAFT_REPO_ROOT/$O/pmc1 -o p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph ) > $O/pmc1.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc $P2 -d $GRAFT_REPO_ROOT/$O/pmc2 -o p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph ) > $O/pmc2.log 2>&1
find $O/pmc1 $O/pmc2 -name "*.db" > $O/dbs.txt
python tools/rocprof_pmc_util.py $O/pmc_util.json "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph; config A" $(cat $O/dbs.txt) > $O/pmc_util.log 2>&1
rm -rf $O/pmc1 $O/pmc2
tail -5 $O/pytest.log
