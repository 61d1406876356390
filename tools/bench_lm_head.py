"""SURVEY section 8f rank 3 / verdict r2 item 9: the LM head - Linear(650 -> 10 000) + log-softmax + NLL (vmlmf_lm.py:355-358,
lm_test.py:140-153) at config E's size (T*B = 8960 rows) - with and without the (T*B, V) score tensor in HBM.

  unfused : torch.addmm (rocBLAS) -> vmlmf_amd.nll_loss (one read forward, one read + one write backward) -> two rocBLAS GEMMs
  chunked : vmlmf_amd.linear_nll(fused=True): rows in chunks, a chunk's scores in one reused cache-resident buffer, the
            backward recomputes them (four GEMM-sized products instead of three, no 358 MB tensor)
Prints one JSON line per (mode, chunk size): forward-only (evaluation) and forward + backward times, hipEvent timed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import vmlmf_amd

T, B, H, V = 35, 256, 650, 10000


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    h = (0.5 * torch.randn(T, B, H, device="cuda")).requires_grad_(True)
    w = (0.05 * torch.randn(V, H, device="cuda")).requires_grad_(True)
    b = torch.zeros(V, device="cuda", requires_grad=True)
    y = torch.randint(0, V, (T, B), device="cuda")
    ref = vmlmf_amd.linear_nll(h, w, b, y, fused=False)
    ref.backward()
    gref = (h.grad.clone(), w.grad.clone(), b.grad.clone())
    flops = 2.0 * T * B * H * V
    for mode, chunk in (("unfused", 0), ("chunked", 256), ("chunked", 512), ("chunked", 1024), ("chunked", 2048)):
        fused = mode == "chunked"

        def fwd():
            with torch.no_grad():
                return vmlmf_amd.linear_nll(h, w, b, y, chunk_rows=chunk or 1024, fused=fused)

        def fwdbwd():
            h.grad = w.grad = b.grad = None
            vmlmf_amd.linear_nll(h, w, b, y, chunk_rows=chunk or 1024, fused=fused).backward()

        loss = float(fwd())
        fwdbwd()
        err = max(float((h.grad - gref[0]).abs().max() / gref[0].abs().max()), float((w.grad - gref[1]).abs().max() / gref[1].abs().max()),
                  float((b.grad - gref[2]).abs().max() / gref[2].abs().max()))
        tf, tfb = timed(fwd), timed(fwdbwd)
        print(json.dumps({"mode": mode, "chunk_rows": chunk, "rows": T * B, "vocab": V, "hidden": H, "loss": round(loss, 5),
                          "loss_ref": round(float(ref), 5), "max_rel_grad_diff_vs_unfused": err, "forward_ms": round(tf, 4),
                          "forward_backward_ms": round(tfb, 4), "forward_tflops": round(flops / tf / 1e9, 1),
                          "score_tensor_MB": round(T * B * V * 4 / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
