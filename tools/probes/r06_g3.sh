set -x
mkdir -p gpurun_out/r06c
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06c/prof32 -o e32 -- python3 $R/tools/probes/rbx_probe.py 32 --stacked-only > $R/gpurun_out/r06c/prof32.log 2>&1
cd $R
DB=$(find gpurun_out/r06c/prof32 -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r06c/e32_stacked_kernel_stats.csv "two PTB group layers at 32 rows, one launch per direction (rbx): rocprofv3 --kernel-trace --stats -- python3 tools/probes/rbx_probe.py 32 --stacked-only"
rm -rf gpurun_out/r06c/prof32
python tools/probes/host_path.py > gpurun_out/r06c/host_path.txt 2>&1
cat gpurun_out/r06c/host_path.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -80 > gpurun_out/r06c/gpu_tests.txt
tail -60 gpurun_out/r06c/gpu_tests.txt
