VMLMF_RBX=2 python tools/probes/rbx_probe.py 32 --one
VMLMF_RBX=2 python tools/probes/rbx_probe.py 128 --one
