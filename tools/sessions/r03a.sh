#!/bin/bash
# r03a: baseline phase stamps of the recurrent kernels at the start of round 3 (rec_probe, x-wave form) + a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
timeout 120 tools/microbench/bin/rec_probe x > $O/probe_xw.txt 2>&1
timeout 120 tools/microbench/bin/rec_probe > $O/probe_loader.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2> $O/bench.err < /dev/null
cat $O/probe_xw.txt $O/probe_loader.txt; cut -c1-600 $O/bench.json
