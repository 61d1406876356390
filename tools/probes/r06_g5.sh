python -m pytest tests/test_gpu_rbx.py tests/test_gpu_stack.py -q -m gpu 2>&1 | tail -2
VMLMF_FFB=1 python -m pytest tests/test_gpu_stack.py -q -m gpu 2>&1 | tail -2
python tools/probes/rbx_probe.py 32 --stacked-only
VMLMF_FFB=0 python tools/probes/rbx_probe.py 32 --stacked-only
python tools/bench_stack.py 2>&1 | tail -3
VMLMF_FFB=1 python tools/bench_stack.py 2>&1 | tail -3
