#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02i; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
python - > $O/c_bf16.jsonl 2> $O/c_bf16.err <<'PY'
import json, sys, time, torch
sys.path.insert(0, ".")
from vmlmf_amd import MyLSTM, MyVMLMFCell, set_compute_dtype, _lib
def graph_of(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
def t(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for B in (128, 1024, 4096):
    for dt, rb in (("f32", 0), ("f32", 1), ("bf16", 1)):
        _lib.tune("rb", -1 if rb == 0 else 1)
        torch.manual_seed(0)
        rnn = MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
        set_compute_dtype(rnn, dt)
        x = torch.randn(B, 24, 77, device="cuda")
        def step():
            rnn.zero_grad(set_to_none=True)
            y, _ = rnn(x); y[:, -1].sum().backward()
        g = graph_of(step)
        ms = t(g.replay, 50) * 1e3
        print(json.dumps({"config": "C: OPP V1 2x256 r=24 T=24", "B": B, "dtype": dt, "kernels": "row-block MFMA" if rb else "VALU row-per-CU", "ms_fwd_bwd": round(ms, 4)}), flush=True)
_lib.tune("rb", -1)
PY
cat $O/c_bf16.jsonl; tail -3 $O/c_bf16.err
