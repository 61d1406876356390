set -x
mkdir -p gpurun_out/r06b
timeout 600 python -m pytest tests/test_gpu_rbx.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r06b/rbx_tests.txt
cat gpurun_out/r06b/rbx_tests.txt
timeout 300 python tools/probes/rbx_probe.py 32 64 128 > gpurun_out/r06b/rbx_probe.txt 2>&1
cat gpurun_out/r06b/rbx_probe.txt
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "packed_image" 2>&1 | tail -40 > gpurun_out/r06b/packed_test.txt
cat gpurun_out/r06b/packed_test.txt
