python -m pytest tests/test_gpu_rbx.py -q -m gpu 2>&1 | grep -E "passed|failed|max err|Error" | head -20
python tools/probes/rbx_probe.py 32 64 128
python tools/probes/rbx_probe.py 32 --plain
python -m pytest tests/test_gpu_rb.py tests/test_dropout.py -q -m gpu -x 2>&1 | tail -2
python tools/bench_rb.py 2>&1 | tail -8
