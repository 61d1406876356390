cp vmlmf_amd/lib/libvmlmf_hip.so /tmp/lib_keep.so
for v in base x0 x01 x012 x3 x34 x01234; do
  cp gpurun_tmp/variants/lib_$v.so vmlmf_amd/lib/libvmlmf_hip.so
  echo "== $v"; python tools/probes/rbx_probe.py 32 --stacked-only 2>&1 | tail -1
done
cp /tmp/lib_keep.so vmlmf_amd/lib/libvmlmf_hip.so
