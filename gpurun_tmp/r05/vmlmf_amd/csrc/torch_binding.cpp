// PyTorch-ROCm binding of the C ABI (include/vmlmf_hip.h) as a TORCH_LIBRARY with C++ autograd functions:
//   vmlmf::sequence        one VMLMF layer over a whole sequence          (replaces the Python time loops vmlmf.py:300-314,
//                                                                          vmlmf_lm.py:272-280 / 166-174 and the cells under them)
//   vmlmf::stack           every layer of a stack in one wavefront launch per direction (the layer loop vmlmf.py:300-314)
//   vmlmf::head_linear     Net.lin on the last timestep                    (vmlmf.py:345,353-355)
//   vmlmf::cross_entropy   the criterion of the reference's training loop  (train.py:58-65)
// PyTorch supplies memory (caching allocator), the current HIP stream and autograd bookkeeping; every float of arithmetic
// happens in libvmlmf_hip.so, reached through the same extern "C" entry points the ctypes binding (vmlmf_amd/_lib.py) calls.
// Why it exists: the unchanged reference loop is eager, and at the UCI-HAR shape the Python autograd.Function bridge spends
// more host time per step (~0.33 ms) than the GPU needs (0.19 ms); in C++ the same bookkeeping costs a fraction of that.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <torch/library.h>
#include <torch/torch.h>

#include <mutex>
#include <tuple>
#include <unordered_map>

#include "../../include/vmlmf_hip.h"

// (PyTorch-ROCm presents HIP devices under the device type "cuda": the guard and stream classes to use are the
// "MasqueradingAsCUDA" ones, the plain c10::hip::HIPGuard refuses a cuda-typed device)
namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc) { TORCH_CHECK(rc == 0, "vmlmf_hip error ", rc, ": ", vmlmf_last_error()); }

void require_hip_f32(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "vmlmf_amd: ", what, " is on ", t.device(),
              "; the VMLMF hot path runs only as HIP kernels on an MI355X (no CPU fallback). Move the module and inputs to 'cuda'.");
  TORCH_CHECK(t.scalar_type() == at::kFloat, "vmlmf_amd: ", what, " must be float32, got ", t.scalar_type());
}

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

const float* cptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* mptr(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// one grow-only scratch buffer per (device, stream): workspace contents never outlive the call that fills them and calls
// on a stream are serialised; under stream capture a private allocation is used (it belongs to the graph's pool)
Tensor workspace(const Tensor& like, size_t nbytes) {
  const auto opts = like.options().dtype(at::kByte);
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(like.device().index());
  if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return at::empty({(int64_t)nbytes}, opts);
  static std::mutex mu;
  static std::unordered_map<uint64_t, Tensor> cache;
  const uint64_t key = ((uint64_t)like.device().index() << 56) ^ (uint64_t)(uintptr_t)stream.stream();
  std::lock_guard<std::mutex> lk(mu);
  Tensor& buf = cache[key];
  if (!buf.defined() || (size_t)buf.numel() < nbytes) buf = at::empty({(int64_t)nbytes}, opts);
  return buf;
}

// ---- parameter lists (order fixed by vmlmf_amd/functional.py) -> vmlmf_params
//   V1-V4: dia_x dia_h u_x v_x b_x b_h u_h[0] v_h[0] (u_h[1] v_h[1])
//   V6:    u_x v_x b_x b_h u_h[0] v_h[0] u_h[1] v_h[1]
//   V5:    w u w1..w4 u1..u4 bias_i bias_f bias_o bias_c
template <class P, class T>
void fill_params(P& p, const std::vector<T>& t, int variant, int g) {
  memset(&p, 0, sizeof(p));
  auto ptr = [&](size_t i) { return (decltype(p.u_x))t[i].template data_ptr<float>(); };
  if (variant == VMLMF_V5_LMF_CELL) {
    p.u_x = ptr(0), p.u_h[0] = ptr(1);
    for (int k = 0; k < 4; ++k) p.w_gate[k] = ptr(2 + k), p.u_gate[k] = ptr(6 + k), p.b_gate[k] = ptr(10 + k);
    return;
  }
  size_t i = 0;
  if (variant != VMLMF_V6_GROUP_NOVM) p.dia_x = ptr(i), p.dia_h = ptr(i + 1), i += 2;
  p.u_x = ptr(i), p.v_x = ptr(i + 1), p.b_x = ptr(i + 2), p.b_h = ptr(i + 3), i += 4;
  for (int s = 0; s < g; ++s) p.u_h[s] = ptr(i + 2 * s), p.v_h[s] = ptr(i + 2 * s + 1);
}

int64_t hidden_size(int variant, const std::vector<Tensor>& params) {
  if (variant == VMLMF_V5_LMF_CELL) return params[1].size(0);          // u (H, ru)
  if (variant == VMLMF_V6_GROUP_NOVM) return params[2].size(-1) / 4;   // bias_x (1, 4H)
  return params[1].size(-1);                                           // dia_h (1, H)
}

vmlmf_desc make_desc(int variant, int64_t B, int64_t T, int64_t I, int64_t H, int64_t w_rank, const std::vector<int64_t>& ur,
                     int64_t g, bool time_major, bool training, int64_t dtype) {
  vmlmf_desc d;
  memset(&d, 0, sizeof(d));
  d.variant = variant, d.B = (int)B, d.T = (int)T, d.I = (int)I, d.H = (int)H, d.w_rank = (int)w_rank, d.g = (int)g;
  for (size_t i = 0; i < VMLMF_MAX_G; ++i) d.u_ranks[i] = i < ur.size() ? (int)ur[i] : 0;
  d.time_major = time_major ? 1 : 0, d.training = training ? 1 : 0, d.dtype = (int)dtype;
  return d;
}

struct SeqFn : public torch::autograd::Function<SeqFn> {
  // args: x, h0 (maybe undefined), c0, params..., then the integer configuration
  // (the parameter list must reach apply() as an at::TensorList: a std::vector would match the "not a tensor" overload
  // of autograd's argument walker and the parameters would get no gradient)
  static variable_list forward(AutogradContext* ctx, Tensor x, c10::optional<Tensor> h0o, c10::optional<Tensor> c0o,
                               at::TensorList params_in, int64_t variant, int64_t g, int64_t w_rank,
                               std::vector<int64_t> u_ranks, bool time_major, bool training, int64_t dtype,
                               c10::optional<Tensor> packed_o, c10::optional<Tensor> head_w_o, c10::optional<Tensor> head_b_o,
                               c10::optional<Tensor> target_o, int64_t ignore_index, c10::optional<Tensor> unit_o,
                               c10::optional<Tensor> ticket_o) {
    // head_w / head_b: a classifier riding on the layer's final hidden state (Net.lin): its logits are the 4th output
    // target (+ the package's unit-gradient tensor and ticket word): the criterion on those logits riding too (vmlmf_ce): its
    // loss is the 5th output, the logits' gradient for d(loss) = 1 is kept for the backward
    Tensor head_w = head_w_o.has_value() ? head_w_o->contiguous() : Tensor();
    Tensor head_b = head_b_o.has_value() ? head_b_o->contiguous() : Tensor();
    if (head_w.defined()) require_hip_f32(head_w, "head weight");
    // packed_o: parameter images kept by the caller (vmlmf_pack_params; functional.PackCache): nothing is packed in this call.
    // It travels as a non-differentiable input and is saved for the backward, which reads the same images.
    Tensor packed = packed_o.has_value() ? *packed_o : Tensor();
    ctx->set_materialize_grads(false);
    require_hip_f32(x, "input");
    x = x.contiguous();
    std::vector<Tensor> params;
    for (const auto& p : params_in) {
      require_hip_f32(p, "parameter");
      params.push_back(p.contiguous());
    }
    Tensor h0 = h0o.has_value() ? h0o->contiguous() : Tensor(), c0 = c0o.has_value() ? c0o->contiguous() : Tensor();
    const int64_t B = time_major ? x.size(1) : x.size(0), T = time_major ? x.size(0) : x.size(1), I = x.size(2);
    const int64_t H = hidden_size((int)variant, params);
    const vmlmf_desc d = make_desc((int)variant, B, T, I, H, w_rank, u_ranks, g, time_major, training, dtype);
    vmlmf_sizes sz;
    check(vmlmf_query(&d, &sz));
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = at::empty(time_major ? at::IntArrayRef({T, B, H}) : at::IntArrayRef({B, T, H}), x.options());
    Tensor hT = at::empty({B, H}, x.options()), cT = at::empty({B, H}, x.options());
    Tensor ws = workspace(x, sz.workspace_bytes);
    Tensor reserve = training ? at::empty({(int64_t)sz.reserve_bytes}, x.options().dtype(at::kByte)) : Tensor();
    vmlmf_params ps;
    fill_params(ps, params, (int)variant, (int)g);
    Tensor logits = head_w.defined() ? at::empty({B, head_w.size(0)}, x.options()) : at::empty({0}, x.options());
    vmlmf_head hd;
    memset(&hd, 0, sizeof(hd));
    if (head_w.defined()) {
      TORCH_CHECK(head_w.dim() == 2 && head_w.size(1) == H, "vmlmf_amd: head weight must be (classes, hidden_size)");
      hd.classes = (int)head_w.size(0), hd.weight = head_w.data_ptr<float>(), hd.bias = cptr(head_b), hd.logits = logits.data_ptr<float>();
    }
    vmlmf_extra ex;
    memset(&ex, 0, sizeof(ex));
    ex.packed = packed.defined() ? packed.data_ptr() : nullptr, ex.head = head_w.defined() ? &hd : nullptr, ex.ce = nullptr;
    Tensor stats, dz_unit, target;
    vmlmf_ce ce;
    memset(&ce, 0, sizeof(ce));
    if (target_o.has_value()) {
      TORCH_CHECK(head_w.defined() && ticket_o.has_value(), "vmlmf_amd: a criterion rides on the classifier's logits (head)");
      target = target_o->contiguous();
      TORCH_CHECK(target.scalar_type() == at::kLong && target.dim() == 1 && target.size(0) == B, "vmlmf_amd: target must be (B,) int64");
      stats = at::empty({2 + B}, x.options());   // loss | nvalid | lse[B]
      if (training) dz_unit = at::empty_like(logits);
      float* sp = stats.data_ptr<float>();
      ce.target = target.data_ptr<int64_t>(), ce.ignore_index = ignore_index, ce.loss = sp, ce.nvalid = sp + 1, ce.lse = sp + 2;
      ce.dlogits_unit = mptr(dz_unit), ce.ticket = (uint64_t*)ticket_o->data_ptr();
      ex.ce = &ce;
    }
    check(vmlmf_seq_forward_ex(&d, &ps, x.data_ptr<float>(), cptr(h0), cptr(c0), y.data_ptr<float>(), hT.data_ptr<float>(),
                               cT.data_ptr<float>(), training ? reserve.data_ptr() : nullptr, ws.data_ptr(), sz.workspace_bytes,
                               stream_of(x), &ex));
    if (training) {
      variable_list saved = {x, y, reserve};
      for (auto& p : params) saved.push_back(p);
      if (h0.defined()) saved.push_back(h0);
      if (c0.defined()) saved.push_back(c0);
      if (packed.defined()) saved.push_back(packed);
      ctx->saved_data["packed"] = packed.defined();
      if (head_w.defined()) saved.push_back(head_w);
      ctx->saved_data["head"] = head_w.defined();
      ctx->saved_data["head_b"] = head_b.defined();
      if (dz_unit.defined()) saved.push_back(dz_unit);
      ctx->saved_data["ce"] = dz_unit.defined();
      ctx->saved_data["unit"] = (unit_o.has_value() && unit_o->defined()) ? (int64_t)(uintptr_t)unit_o->data_ptr() : (int64_t)0;
      ctx->save_for_backward(saved);
      ctx->saved_data["np"] = (int64_t)params.size();
      ctx->saved_data["h0"] = h0.defined();
      ctx->saved_data["c0"] = c0.defined();
      ctx->saved_data["cfg"] = std::vector<int64_t>{variant, g, w_rank, time_major ? 1 : 0, B, T, I, H, dtype};
      ctx->saved_data["ur"] = u_ranks;
    }
    return {y, hT, cT, logits, stats.defined() ? stats.select(0, 0) : at::empty({0}, x.options())};
  }

  static variable_list backward(AutogradContext* ctx, variable_list gout) {
    const auto saved = ctx->get_saved_variables();
    const int64_t np = ctx->saved_data["np"].toInt();
    const bool has_h0 = ctx->saved_data["h0"].toBool(), has_c0 = ctx->saved_data["c0"].toBool();
    const auto cfg = ctx->saved_data["cfg"].toIntVector();
    const auto ur = ctx->saved_data["ur"].toIntVector();
    const int64_t variant = cfg[0], g = cfg[1], w_rank = cfg[2], B = cfg[4], T = cfg[5], I = cfg[6], H = cfg[7];
    const bool time_major = cfg[3] != 0;
    const Tensor &x = saved[0], &y = saved[1], &reserve = saved[2];
    std::vector<Tensor> params(saved.begin() + 3, saved.begin() + 3 + np);
    size_t k = 3 + np;
    Tensor h0 = has_h0 ? saved[k++] : Tensor(), c0 = has_c0 ? saved[k++] : Tensor();
    Tensor packed = ctx->saved_data["packed"].toBool() ? saved[k++] : Tensor();
    Tensor head_w = ctx->saved_data["head"].toBool() ? saved[k++] : Tensor();
    const bool has_head_b = ctx->saved_data["head_b"].toBool();
    Tensor dlogits = (head_w.defined() && gout.size() > 3 && gout[3].defined()) ? gout[3].contiguous() : Tensor();
    if (ctx->saved_data["ce"].toBool() && gout.size() > 4 && gout[4].defined()) {
      // the criterion's share of d(logits): what the forward launch wrote for d(loss) = 1 - as it is when the incoming gradient
      // IS the package's constant one, scaled otherwise
      Tensor dz = saved[k++];
      const int64_t unit = ctx->saved_data["unit"].toInt();
      if (!(unit != 0 && (int64_t)(uintptr_t)gout[4].data_ptr() == unit)) dz = dz * gout[4];
      dlogits = dlogits.defined() ? dlogits + dz : dz;
    }
    Tensor dy = gout[0].defined() ? gout[0].contiguous() : Tensor();
    Tensor dhT = gout[1].defined() ? gout[1].contiguous() : Tensor();
    Tensor dcT = gout[2].defined() ? gout[2].contiguous() : Tensor();
    const vmlmf_desc d = make_desc((int)variant, B, T, I, H, w_rank, ur, g, time_major, true, cfg[8]);
    vmlmf_sizes sz;
    check(vmlmf_query(&d, &sz));
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor dx = ctx->needs_input_grad(0) ? at::empty_like(x) : Tensor();
    Tensor dh0 = has_h0 ? at::empty({B, H}, x.options()) : Tensor(), dc0 = has_c0 ? at::empty({B, H}, x.options()) : Tensor();
    // one flat buffer for all parameter gradients (views are returned): a single allocation, contiguous for the
    // data-parallel all-reduce (vmlmf_amd/dp.py)
    // (the classifier's weight and bias gradients are the tail of the same allocation: ONE flat buffer, ONE all-reduce per step,
    // SURVEY section 8e)
    int64_t total = 0;
    for (const auto& p : params) total += p.numel();
    const int64_t head_floats = dlogits.defined() ? head_w.size(0) * H + head_w.size(0) : 0;
    Tensor flat = at::empty({total + head_floats}, x.options());
    std::vector<Tensor> grads;
    int64_t o = 0;
    for (const auto& p : params) {
      grads.push_back(flat.as_strided(p.sizes(), p.strides(), o));   // one op per view (narrow + view were two)
      o += p.numel();
    }
    Tensor ws = workspace(x, sz.workspace_bytes);
    vmlmf_params ps;
    vmlmf_grads gs;
    fill_params(ps, params, (int)variant, (int)g);
    fill_params(gs, grads, (int)variant, (int)g);
    // classifier gradients: behind the layer's in the same allocation
    Tensor dW, db;
    vmlmf_head hd;
    memset(&hd, 0, sizeof(hd));
    if (dlogits.defined()) {
      const int64_t C = head_w.size(0);
      dW = flat.narrow(0, total, C * H).view({C, H});
      if (has_head_b) db = flat.narrow(0, total + C * H, C);
      hd.classes = (int)C, hd.weight = head_w.data_ptr<float>(), hd.dlogits = dlogits.data_ptr<float>();
      hd.dweight = dW.data_ptr<float>(), hd.dbias = has_head_b ? db.data_ptr<float>() : nullptr;
    }
    vmlmf_extra ex;
    memset(&ex, 0, sizeof(ex));
    ex.packed = packed.defined() ? packed.data_ptr() : nullptr, ex.head = dlogits.defined() ? &hd : nullptr, ex.ce = nullptr;
    check(vmlmf_seq_backward_ex(&d, &ps, x.data_ptr<float>(), cptr(h0), cptr(c0), y.data_ptr<float>(), reserve.data_ptr(),
                                cptr(dy), cptr(dhT), cptr(dcT), mptr(dx), mptr(dh0), mptr(dc0), &gs, ws.data_ptr(),
                                sz.workspace_bytes, stream_of(x), &ex));
    variable_list out = {dx, dh0, dc0};
    for (auto& gt : grads) out.push_back(gt);
    for (int i = 0; i < 8; ++i) out.push_back(Tensor());   // the integer configuration and the kept parameter images
    out.push_back(dW);                                     // head weight, head bias
    out.push_back(db);
    for (int i = 0; i < 4; ++i) out.push_back(Tensor());   // target, ignore_index, unit, ticket
    return out;
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> sequence(const Tensor& x, const c10::optional<Tensor>& h0, const c10::optional<Tensor>& c0,
                                            at::TensorList params, int64_t variant, int64_t g, int64_t w_rank,
                                            at::IntArrayRef u_ranks, bool time_major, int64_t dtype,
                                            const c10::optional<Tensor>& packed, const c10::optional<Tensor>& head_w,
                                            const c10::optional<Tensor>& head_b) {
  // grad mode is off inside Function::forward: whether the tape is needed is decided here (False under torch.no_grad():
  // inference kernels, no reserve buffer)
  bool training = x.requires_grad() || (h0.has_value() && h0->requires_grad()) || (c0.has_value() && c0->requires_grad());
  for (const auto& p : params) training = training || p.requires_grad();
  training = training || (head_w.has_value() && head_w->requires_grad()) || (head_b.has_value() && head_b->requires_grad());
  training = training && at::GradMode::is_enabled();
  auto out = SeqFn::apply(x, h0, c0, params, variant, g, w_rank, u_ranks.vec(), time_major, training, dtype, packed, head_w, head_b,
                          c10::optional<Tensor>(), (int64_t)-100, c10::optional<Tensor>(), c10::optional<Tensor>());
  return {out[0], out[1], out[2], out[3]};
}

// the same layer with the criterion of the reference's loop riding on its classifier (vmlmf_ce): y, hT, cT, logits, loss
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> sequence_loss(const Tensor& x, const c10::optional<Tensor>& h0,
                                                                 const c10::optional<Tensor>& c0, at::TensorList params, int64_t variant,
                                                                 int64_t g, int64_t w_rank, at::IntArrayRef u_ranks, bool time_major,
                                                                 int64_t dtype, const c10::optional<Tensor>& packed, const Tensor& head_w,
                                                                 const c10::optional<Tensor>& head_b, const Tensor& target,
                                                                 int64_t ignore_index, const Tensor& unit, const Tensor& ticket) {
  bool training = x.requires_grad() || (h0.has_value() && h0->requires_grad()) || (c0.has_value() && c0->requires_grad());
  for (const auto& p : params) training = training || p.requires_grad();
  training = training || head_w.requires_grad() || (head_b.has_value() && head_b->requires_grad());
  training = training && at::GradMode::is_enabled();
  auto out = SeqFn::apply(x, h0, c0, params, variant, g, w_rank, u_ranks.vec(), time_major, training, dtype, packed,
                          c10::optional<Tensor>(head_w), head_b, c10::optional<Tensor>(target), ignore_index, c10::optional<Tensor>(unit),
                          c10::optional<Tensor>(ticket));
  return {out[0], out[1], out[2], out[3], out[4]};
}

// ---- stacked layers: one wavefront launch per direction (C ABI 7: vmlmf_stack_*) ------------------------------------
// outputs: y of the top layer, hT and cT as (L, B, H) tensors, the logits of a classifier riding on the top layer (or an
// empty tensor).  Initial states are zero (MyLSTM.forward).
struct StackFn : public torch::autograd::Function<StackFn> {
  static variable_list forward(AutogradContext* ctx, Tensor x, at::TensorList params_in, int64_t L, int64_t variant, int64_t w_rank,
                               std::vector<int64_t> u_ranks, int64_t g, bool time_major, bool training,
                               c10::optional<Tensor> head_w_o, c10::optional<Tensor> head_b_o) {
    Tensor head_w = head_w_o.has_value() ? head_w_o->contiguous() : Tensor();
    Tensor head_b = head_b_o.has_value() ? head_b_o->contiguous() : Tensor();
    if (head_w.defined()) require_hip_f32(head_w, "head weight");
    ctx->set_materialize_grads(false);
    require_hip_f32(x, "input");
    x = x.contiguous();
    std::vector<Tensor> params;
    for (const auto& p : params_in) {
      require_hip_f32(p, "parameter");
      params.push_back(p.contiguous());
    }
    TORCH_CHECK(L >= 1 && L <= VMLMF_STACK_MAX && params.size() % L == 0, "vmlmf_amd: bad stack");
    const size_t nper = params.size() / L;
    const int64_t B = time_major ? x.size(1) : x.size(0), T = time_major ? x.size(0) : x.size(1), I = x.size(2);
    std::vector<Tensor> p0(params.begin(), params.begin() + nper);
    const int64_t H = hidden_size((int)variant, p0);
    std::vector<vmlmf_stack_layer> ly(L);
    std::vector<vmlmf_params> ps(L);
    std::vector<size_t> rbytes(L);
    size_t wbytes = 0;
    memset(ly.data(), 0, sizeof(vmlmf_stack_layer) * L);
    for (int64_t l = 0; l < L; ++l)
      ly[l].desc = make_desc((int)variant, B, T, l == 0 ? I : H, H, w_rank, u_ranks, g, time_major, training, VMLMF_DT_F32);
    check(vmlmf_stack_query((int)L, ly.data(), rbytes.data(), &wbytes));
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    std::vector<Tensor> ys, reserves;
    Tensor hT = at::empty({L, B, H}, x.options()), cT = at::empty({L, B, H}, x.options());
    Tensor ws = workspace(x, wbytes);
    for (int64_t l = 0; l < L; ++l) {
      ys.push_back(at::empty(time_major ? at::IntArrayRef({T, B, H}) : at::IntArrayRef({B, T, H}), x.options()));
      reserves.push_back(training ? at::empty({(int64_t)rbytes[l]}, x.options().dtype(at::kByte)) : Tensor());
      std::vector<Tensor> pl(params.begin() + l * nper, params.begin() + (l + 1) * nper);
      fill_params(ps[l], pl, (int)variant, (int)g);
      ly[l].params = &ps[l];
      ly[l].y = ys[l].data_ptr<float>(), ly[l].hT = hT.data_ptr<float>() + l * B * H, ly[l].cT = cT.data_ptr<float>() + l * B * H;
      ly[l].reserve = training ? reserves[l].data_ptr() : nullptr;
    }
    Tensor logits = head_w.defined() ? at::empty({B, head_w.size(0)}, x.options()) : at::empty({0}, x.options());
    vmlmf_head hd;
    memset(&hd, 0, sizeof(hd));
    if (head_w.defined()) {
      TORCH_CHECK(head_w.dim() == 2 && head_w.size(1) == H, "vmlmf_amd: head weight must be (classes, hidden_size)");
      hd.classes = (int)head_w.size(0), hd.weight = head_w.data_ptr<float>(), hd.bias = cptr(head_b), hd.logits = logits.data_ptr<float>();
    }
    check(vmlmf_stack_forward((int)L, ly.data(), x.data_ptr<float>(), head_w.defined() ? &hd : nullptr, ws.data_ptr(), wbytes,
                              stream_of(x)));
    if (training) {
      variable_list saved = {x};
      for (auto& t : ys) saved.push_back(t);
      for (auto& t : reserves) saved.push_back(t);
      for (auto& t : params) saved.push_back(t);
      if (head_w.defined()) saved.push_back(head_w);
      ctx->save_for_backward(saved);
      ctx->saved_data["cfg"] = std::vector<int64_t>{L, variant, w_rank, g, time_major ? 1 : 0, B, T, I, H, (int64_t)nper};
      ctx->saved_data["ur"] = u_ranks;
      ctx->saved_data["head"] = head_w.defined();
      ctx->saved_data["head_b"] = head_b.defined();
    }
    return {ys[L - 1], hT, cT, logits};
  }

  static variable_list backward(AutogradContext* ctx, variable_list gout) {
    const auto saved = ctx->get_saved_variables();
    const auto cfg = ctx->saved_data["cfg"].toIntVector();
    const int64_t L = cfg[0], variant = cfg[1], w_rank = cfg[2], g = cfg[3], B = cfg[5], T = cfg[6], I = cfg[7], H = cfg[8], nper = cfg[9];
    const auto u_ranks = ctx->saved_data["ur"].toIntVector();
    const bool time_major = cfg[4] != 0;
    const Tensor& x = saved[0];
    const bool has_head = ctx->saved_data["head"].toBool(), has_head_b = ctx->saved_data["head_b"].toBool();
    std::vector<Tensor> params(saved.begin() + 1 + 2 * L, saved.end() - (has_head ? 1 : 0));
    Tensor head_w = has_head ? saved.back() : Tensor();
    Tensor dlogits = (has_head && gout.size() > 3 && gout[3].defined()) ? gout[3].contiguous() : Tensor();
    Tensor dy = gout[0].defined() ? gout[0].contiguous() : Tensor();
    Tensor dhT = gout[1].defined() ? gout[1].contiguous() : Tensor();
    Tensor dcT = gout[2].defined() ? gout[2].contiguous() : Tensor();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor dx = ctx->needs_input_grad(0) ? at::empty_like(x) : Tensor();
    int64_t total = 0;
    for (const auto& p : params) total += p.numel();
    const int64_t head_floats = dlogits.defined() ? head_w.size(0) * H + head_w.size(0) : 0;
    // the parameter gradients of the whole stack AND of the classifier in one allocation (views are returned): one all-reduce
    Tensor flat = at::empty({total + head_floats}, x.options());
    std::vector<Tensor> grads;
    int64_t o = 0;
    for (const auto& p : params) {
      grads.push_back(flat.as_strided(p.sizes(), p.strides(), o));   // one op per view (narrow + view were two)
      o += p.numel();
    }
    std::vector<vmlmf_stack_layer> ly(L);
    std::vector<vmlmf_params> ps(L);
    std::vector<vmlmf_grads> gs(L);
    std::vector<size_t> rbytes(L);
    size_t wbytes = 0;
    memset(ly.data(), 0, sizeof(vmlmf_stack_layer) * L);
    for (int64_t l = 0; l < L; ++l) {
      ly[l].desc = make_desc((int)variant, B, T, l == 0 ? I : H, H, w_rank, u_ranks, g, time_major, true, VMLMF_DT_F32);
      std::vector<Tensor> pl(params.begin() + l * nper, params.begin() + (l + 1) * nper);
      std::vector<Tensor> gl(grads.begin() + l * nper, grads.begin() + (l + 1) * nper);
      fill_params(ps[l], pl, (int)variant, (int)g);
      fill_params(gs[l], gl, (int)variant, (int)g);
      ly[l].params = &ps[l], ly[l].grads = &gs[l];
      ly[l].y = const_cast<float*>(saved[1 + l].data_ptr<float>());
      ly[l].reserve = saved[1 + L + l].data_ptr();
      ly[l].dhT = dhT.defined() ? dhT.data_ptr<float>() + l * B * H : nullptr;
      ly[l].dcT = dcT.defined() ? dcT.data_ptr<float>() + l * B * H : nullptr;
    }
    check(vmlmf_stack_query((int)L, ly.data(), rbytes.data(), &wbytes));
    Tensor ws = workspace(x, wbytes);
    Tensor dW, db;
    vmlmf_head hd;
    memset(&hd, 0, sizeof(hd));
    if (dlogits.defined()) {   // classifier gradients: the tail of the stack's allocation
      const int64_t C = head_w.size(0);
      dW = flat.narrow(0, total, C * H).view({C, H});
      if (has_head_b) db = flat.narrow(0, total + C * H, C);
      hd.classes = (int)C, hd.weight = head_w.data_ptr<float>(), hd.dlogits = dlogits.data_ptr<float>();
      hd.dweight = dW.data_ptr<float>(), hd.dbias = has_head_b ? db.data_ptr<float>() : nullptr;
    }
    check(vmlmf_stack_backward((int)L, ly.data(), x.data_ptr<float>(), cptr(dy), mptr(dx), dlogits.defined() ? &hd : nullptr,
                               ws.data_ptr(), wbytes, stream_of(x)));
    variable_list out = {dx};
    for (auto& gt : grads) out.push_back(gt);
    for (int i = 0; i < 7; ++i) out.push_back(Tensor());   // the integer configuration
    out.push_back(dW);                                     // head weight, head bias
    out.push_back(db);
    return out;
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> stack(const Tensor& x, at::TensorList params, int64_t L, int64_t variant, int64_t w_rank,
                                                 at::IntArrayRef u_ranks, int64_t g, bool time_major,
                                                 const c10::optional<Tensor>& head_w, const c10::optional<Tensor>& head_b) {
  bool training = x.requires_grad();
  for (const auto& p : params) training = training || p.requires_grad();
  training = training || (head_w.has_value() && head_w->requires_grad()) || (head_b.has_value() && head_b->requires_grad());
  training = training && at::GradMode::is_enabled();
  auto out = StackFn::apply(x, params, L, variant, w_rank, u_ranks.vec(), g, time_major, training, head_w, head_b);
  return {out[0], out[1], out[2], out[3]};
}

// ---- classifier head ---------------------------------------------------------------------------------------------
struct HeadFn : public torch::autograd::Function<HeadFn> {
  static Tensor forward(AutogradContext* ctx, Tensor h, Tensor weight, c10::optional<Tensor> bias) {
    ctx->set_materialize_grads(false);
    require_hip_f32(h, "head input");
    require_hip_f32(weight, "head weight");
    if (h.stride(-1) != 1) h = h.contiguous();
    weight = weight.contiguous();
    Tensor b = bias.has_value() ? bias->contiguous() : Tensor();
    const int64_t B = h.size(0), H = h.size(1), C = weight.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(h.device());
    Tensor out = at::empty({B, C}, h.options());
    check(vmlmf_head_forward((int)B, (int)H, (int)C, h.data_ptr<float>(), h.stride(0), weight.data_ptr<float>(), cptr(b),
                             out.data_ptr<float>(), stream_of(h)));
    ctx->save_for_backward({h, weight});
    ctx->saved_data["bias"] = b.defined();
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gout) {
    if (!gout[0].defined()) return {Tensor(), Tensor(), Tensor()};
    const auto saved = ctx->get_saved_variables();
    const Tensor &h = saved[0], &weight = saved[1];
    Tensor dl = gout[0].contiguous();
    const int64_t B = h.size(0), H = h.size(1), C = weight.size(0);
    const bool need_h = ctx->needs_input_grad(0), need_w = ctx->needs_input_grad(1);
    const bool need_b = ctx->saved_data["bias"].toBool() && ctx->needs_input_grad(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(h.device());
    Tensor dh = need_h ? at::empty({B, H}, h.options()) : Tensor();
    Tensor flat = (need_w || need_b) ? at::empty({C * H + C}, h.options()) : Tensor();   // weight + bias gradients share one allocation
    Tensor dW = need_w ? flat.narrow(0, 0, C * H).view({C, H}) : Tensor();
    Tensor db = need_b ? flat.narrow(0, C * H, C) : Tensor();
    check(vmlmf_head_backward((int)B, (int)H, (int)C, h.data_ptr<float>(), h.stride(0), weight.data_ptr<float>(),
                              dl.data_ptr<float>(), mptr(dh), mptr(dW), mptr(db), stream_of(h)));
    return {dh, dW, db};
  }
};

Tensor head_linear(const Tensor& h, const Tensor& weight, const c10::optional<Tensor>& bias) {
  return HeadFn::apply(h, weight, bias);
}

// ---- cross-entropy -----------------------------------------------------------------------------------------------
struct CeFn : public torch::autograd::Function<CeFn> {
  // `unit`: the package's constant d(loss) = 1 tensor (vmlmf_amd.unit_gradient); a backward that is handed exactly that
  // tensor returns the gradient the forward kernel already wrote, without a launch
  static Tensor forward(AutogradContext* ctx, Tensor logits, Tensor target, int64_t ignore_index, Tensor unit, bool need) {
    require_hip_f32(logits, "logits");
    const bool leaf = !logits.grad_fn();
    logits = logits.contiguous();
    target = target.contiguous();
    const int64_t B = logits.size(0), C = logits.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    Tensor stats = at::empty({B + 2}, logits.options());   // loss | nvalid | lse[B]
    Tensor dz = need ? at::empty_like(logits) : Tensor();
    float* sp = stats.data_ptr<float>();
    check(vmlmf_ce_forward((int)B, (int)C, logits.data_ptr<float>(), target.data_ptr<int64_t>(), ignore_index, sp, sp + 2, sp + 1,
                           mptr(dz), stream_of(logits)));
    variable_list saved = {logits, target, stats};
    if (dz.defined()) saved.push_back(dz);
    ctx->save_for_backward(saved);
    ctx->saved_data["ignore"] = ignore_index;
    ctx->saved_data["leaf"] = leaf;
    ctx->saved_data["unit"] = unit.defined() ? (int64_t)(uintptr_t)unit.data_ptr() : (int64_t)0;
    return stats.select(0, 0);
  }
  static variable_list backward(AutogradContext* ctx, variable_list gout) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &logits = saved[0], &target = saved[1], &stats = saved[2];
    const int64_t unit = ctx->saved_data["unit"].toInt();
    if (saved.size() == 4 && unit != 0 && (int64_t)(uintptr_t)gout[0].data_ptr() == unit) {
      // leaf logits: AccumulateGrad adopts what it is handed, and in-place work on that .grad must not reach the saved buffer
      return {ctx->saved_data["leaf"].toBool() ? saved[3].clone() : saved[3], Tensor(), Tensor(), Tensor(), Tensor()};
    }
    const int64_t B = logits.size(0), C = logits.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(logits.device());
    Tensor dloss = gout[0].contiguous();
    Tensor dz = at::empty_like(logits);
    const float* sp = stats.data_ptr<float>();
    check(vmlmf_ce_backward((int)B, (int)C, logits.data_ptr<float>(), target.data_ptr<int64_t>(), ctx->saved_data["ignore"].toInt(),
                            sp + 2, sp + 1, dloss.data_ptr<float>(), dz.data_ptr<float>(), stream_of(logits)));
    return {dz, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor cross_entropy(const Tensor& logits, const Tensor& target, int64_t ignore_index, const Tensor& unit) {
  return CeFn::apply(logits, target, ignore_index, unit, logits.requires_grad() && at::GradMode::is_enabled());
}

}  // namespace

TORCH_LIBRARY(vmlmf, m) {
  m.def("sequence(Tensor x, Tensor? h0, Tensor? c0, Tensor[] params, int variant, int g, int w_rank, int[] u_ranks, bool time_major, int dtype, Tensor? packed, Tensor? head_w, Tensor? head_b) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("sequence_loss(Tensor x, Tensor? h0, Tensor? c0, Tensor[] params, int variant, int g, int w_rank, int[] u_ranks, bool time_major, int dtype, Tensor? packed, Tensor head_w, Tensor? head_b, Tensor target, int ignore_index, Tensor unit, Tensor ticket) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("stack(Tensor x, Tensor[] params, int L, int variant, int w_rank, int[] u_ranks, int g, bool time_major, Tensor? head_w, Tensor? head_b) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("head_linear(Tensor h, Tensor weight, Tensor? bias) -> Tensor");
  m.def("cross_entropy(Tensor logits, Tensor target, int ignore_index, Tensor unit) -> Tensor");
}

// registered for every dispatch key that reaches them: the functions build their own autograd nodes
TORCH_LIBRARY_IMPL(vmlmf, CompositeImplicitAutograd, m) {
  m.impl("sequence", sequence);
  m.impl("sequence_loss", sequence_loss);
  m.impl("stack", stack);
  m.impl("head_linear", head_linear);
  m.impl("cross_entropy", cross_entropy);
}
