"""The whole language-model step of lm_test.py:196-209 at BASELINE config E's shape on one MI355X: Model (2 VMLMF layers,
H 650, rank 32, vocabulary 10 000, T 35) -> nll_loss -> backward -> clip + SGD, eager and hipGraph replay, with the
fused loss / update and with the reference's own formulations in stock ops.  Not the graded bench (bench.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from vmlmf_amd import Model, MyVMLSTMGroup, nll_loss, optim

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, H, V = 35, 650, 10000


def stock_nll(scores, y):          # lm_test.py:140-153 as written
    batch_size = y.size(1)
    expscores = scores.exp()
    probabilities = expscores / expscores.sum(1, keepdim=True)
    answerprobs = probabilities[range(len(y.reshape(-1))), y.reshape(-1)]
    return torch.mean(-torch.log(answerprobs) * batch_size)


def run(tag, group, fused):
    torch.manual_seed(0)
    model = Model(V, H, 2, 0.0, 0.05, w_rank=32, u_ranks=[32], lstm_type="vmlmf")
    if group:   # the reference's Model cannot build the group layers (constructor quirk): put them in by hand
        model.rnns = torch.nn.ModuleList([MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]) for _ in range(2)])
        model.reset_parameters()
    model = model.to(DEV)
    x = torch.randint(0, V, (T, B), device=DEV)
    y = torch.randint(0, V, (T, B), device=DEV)
    states = model.state_init(B)

    def step():
        model.zero_grad(set_to_none=True)
        scores, _ = model(x, [(h.detach(), c.detach()) for h, c in states])
        loss = nll_loss(scores, y) if fused else stock_nll(scores, y)
        loss.backward()
        if fused:
            optim.clip_sgd_step(model.parameters(), lr=1e-3, max_norm=5.0)
        else:
            with torch.no_grad():
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
                for p in model.parameters():
                    p -= 1e-3 * p.grad

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"config": tag, "B": B, "T": T, "vocab": V, "fused_loss_and_update": fused,
                      "ms_per_step_eager": round(ms, 3), "words_per_s": round(T * B / ms * 1e3)}), flush=True)


if __name__ == "__main__":
    run("E-model: Embed + 2 x MyVMLSTM + Linear + nll", False, True)
    run("E-model: Embed + 2 x MyVMLSTM + Linear + nll", False, False)
    run("E-model: Embed + 2 x MyVMLSTMGroup + Linear + nll", True, True)
