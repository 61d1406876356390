#!/bin/bash
# GPU session r02a: round-2 baseline. tests, bench (1 GPU, forced collective, strong mode), counter list, PMC passes, configs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --force-collective --no-cpu-baseline --steps 100 > $O/bench_fc.json 2> $O/bench_fc.err
python bench.py --global-batch 512 --no-cpu-baseline --steps 100 > $O/bench_b512.json 2> $O/bench_b512.err
rocprofv3 -L > $O/counters.txt 2>&1

P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
( cd /tmp && rocprofv3 --kernel-trace --pmc $P1 -d $GRAFT_REPO_ROOT/$O/pmc1 -o p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph ) > $O/pmc1.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc $P2 -d $GRAFT_REPO_ROOT/$O/pmc2 -o p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph ) > $O/pmc2.log 2>&1
find $O/pmc1 $O/pmc2 -name "*.db" > $O/dbs.txt
python tools/rocprof_pmc_util.py $O/pmc_util.json "r02a" $(cat $O/dbs.txt) > $O/pmc_util.log 2>&1
python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
ls -la $O
