python -m pytest tests/test_gpu_rbx.py -q -m gpu 2>&1 | tail -2
cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for v in keep f2b2 f4b6 f2b3 f4b2; do
  if [ $v = keep ]; then cp /tmp/lib_keep.so vmlmf_amd/lib/libvmlmf_hip.so; else cp gpurun_tmp/variants/lib_$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  echo "== $v"; python tools/probes/rbx_probe.py 32 --stacked-only 2>&1 | tail -1
done; done
cp /tmp/lib_keep.so vmlmf_amd/lib/libvmlmf_hip.so
python tools/probes/rbx_probe.py 32 --stacked-only --plain 2>&1 | tail -1
