#!/bin/bash
# r03b: the storer as wave NW+4 (on the x-wave's / loader's SIMD) against wave NW+1 (on compute wave 0's SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
for v in rec_probe rec_probe_stw7; do
  timeout 120 tools/microbench/bin/$v x > $O/${v}_xw.txt 2>&1
  timeout 120 tools/microbench/bin/$v > $O/${v}_loader.txt 2>&1
  echo "== $v"; cat $O/${v}_xw.txt $O/${v}_loader.txt
done
