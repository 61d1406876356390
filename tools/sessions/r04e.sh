#!/bin/bash
# r04e: the LM head for training (loss + gradient in place, tuned GEMM forms), the embedding gradient kernel: parity, then the LM step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "fused_head or nll_forward_grad or embedding_gradient or lm_network" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -25 $O/tests.txt
timeout 600 python tools/bench_lm.py > $O/lm.jsonl 2> $O/lm.err; cat $O/lm.jsonl; tail -3 $O/lm.err
timeout 600 python tools/bench_lm.py 32 > $O/lm_b32.jsonl 2>> $O/lm.err; cat $O/lm_b32.jsonl
python - <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from vmlmf_amd.functional import head_forms
f = head_forms(8960, 650, 10000, torch.device('cuda', 0))
print({k: (v[0], v[1], round(v[3], 3) if v[3] else None) for k, v in f.items()})
PY
