#!/bin/bash
# r04a: round-4 opening state on the box: GPU tests, the bench line, fresh per-kernel splits of the LM step and of config E's layers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
timeout 300 python tools/bench_lm.py > $O/lm.jsonl 2> $O/lm.err; cat $O/lm.jsonl
timeout 300 python tools/bench_lm.py 32 > $O/lm_b32.jsonl 2>> $O/lm.err; cat $O/lm_b32.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lm -o lm -- python3 $GRAFT_REPO_ROOT/tools/bench_lm.py > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_lm.err
DB=$(ls /tmp/prof_lm/*.db /tmp/prof_lm/*/*.db 2>/dev/null | head -1); echo "db=$DB"
cd $GRAFT_REPO_ROOT
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/lm_step_kernel_stats.csv "tools/bench_lm.py: whole LM steps at config E's shape (13 steps each of fused / stock / group)" > /dev/null
head -45 $O/lm_step_kernel_stats.csv | cut -c1-160
