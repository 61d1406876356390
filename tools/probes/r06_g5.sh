python -m pytest tests/test_gpu_stack.py tests/test_gpu_rbx.py tests/test_dropout.py -q -m gpu -x 2>&1 | tail -2
python tools/probes/rbx_probe.py 32 64 128
python tools/probes/rbx_probe.py 32 --plain
for i in 1 2; do
  (cd gpurun_tmp/r05 && python tools/bench_configs.py 2>/dev/null | grep "C(fp32)" | cut -c1-140 | sed 's/^/r05 /')
  python tools/bench_configs.py 2>/dev/null | grep "C(fp32)" | cut -c1-140 | sed 's/^/r06 /'
done
