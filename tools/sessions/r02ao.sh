#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_rb.py e_rows 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ao_e_rows.log
