# round 6: stacks of layers with different hidden sizes - parity tests, then config C same-box A/B against the round-5 tree
set -x
mkdir -p gpurun_out/r06u
python -m pytest tests/test_gpu_stack.py tests/test_dropout.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06u/t_stack.txt; cat gpurun_out/r06u/t_stack.txt
python -m pytest tests/test_gpu_modules.py tests/test_gpu_rbx.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06u/t_mod.txt; cat gpurun_out/r06u/t_mod.txt
bash tools/probes/r06_ab_c.sh > gpurun_out/r06u/ab_c.txt 2>&1; cat gpurun_out/r06u/ab_c.txt
