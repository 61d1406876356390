cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/lib_keep.so
for v in keep a1 a2 a3; do
  if [ $v = keep ]; then cp /tmp/lib_keep.so vmlmf_amd/lib/libvmlmf_hip.so; else cp gpurun_tmp/variants/lib_$v.so vmlmf_amd/lib/libvmlmf_hip.so; fi
  echo "== $v"; VMLMF_RBX=2 python tools/probes/rbx_probe.py 32 --one --stacked-only 2>&1 | tail -1; python tools/probes/rbx_probe.py 32 --stacked-only 2>&1 | tail -1
done
cp /tmp/lib_keep.so vmlmf_amd/lib/libvmlmf_hip.so
