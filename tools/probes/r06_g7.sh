mkdir -p gpurun_out/r06g
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06g/gpu_tests.txt; tail -4 gpurun_out/r06g/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err; tail -c 600 gpurun_out/r06g/bench.err
python tools/bench_lm.py --help 2>&1 | head -30
