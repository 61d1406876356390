set -x
mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_rbx.py -x -q -m gpu 2>&1 | tail -5
python tools/probes/rbx_probe.py 32 > gpurun_out/r06d/p_base.txt 2>&1; cat gpurun_out/r06d/p_base.txt
VMLMF_RBX_DBG=1 python tools/probes/rbx_probe.py 32 --stacked-only > gpurun_out/r06d/p_dbg1.txt 2>&1; cat gpurun_out/r06d/p_dbg1.txt
VMLMF_RBX_DBG=5 python tools/probes/rbx_probe.py 32 --stacked-only > gpurun_out/r06d/p_dbg5.txt 2>&1; cat gpurun_out/r06d/p_dbg5.txt
VMLMF_RBX=2 python tools/probes/rbx_probe.py 32 --one > gpurun_out/r06d/p_one.txt 2>&1; cat gpurun_out/r06d/p_one.txt
VMLMF_RBX=2 python tools/probes/rbx_probe.py 32 --one --plain > gpurun_out/r06d/p_one_plain.txt 2>&1; cat gpurun_out/r06d/p_one_plain.txt
python tools/probes/rbx_probe.py 32 --plain > gpurun_out/r06d/p_plain.txt 2>&1; cat gpurun_out/r06d/p_plain.txt
