set -x
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "dirty or graded or time_major or graphed or criterion" 2>&1 | tail -15 > gpurun_out/r06a/new_tests.txt
python tools/probes/host_path.py --prof > gpurun_out/r06a/host_path.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06a/gpu_tests.txt
cat gpurun_out/r06a/new_tests.txt gpurun_out/r06a/host_path.txt
tail -3 gpurun_out/r06a/gpu_tests.txt
