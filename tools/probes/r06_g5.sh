python -m pytest tests/test_gpu_rbx.py -q -m gpu 2>&1 | tail -3
python tools/probes/rbx_probe.py 32 64 128
python tools/probes/rbx_probe.py 32 --plain
python -m pytest tests/test_dropout.py tests/test_gpu_modules.py -q -m gpu -x 2>&1 | tail -3
