// Internal launcher interface between the API translation unit and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "vmlmf_device.h"
#include "vmlmf_dropout.h"

// launch bounds of rec_fwd_kernel / rec_bwd_kernel.  (The storer as wave NW + 4, i.e. on the SIMD of the loader / x-projection
// wave instead of compute wave 0's: -1.5 us with the loader, +1.8 us with the x-projection wave, not shipped -
// tools/experiments/ablation_switches.patch, -DVMLMF_STW7.)
#define VG_REC_BOUNDS (MAXT + 128)

// Classifier riding on a layer (Net.lin on the final hidden state, vmlmf.py:345,353-355): logits in the epilogue of the
// forward recurrence, d(hT) = dlogits W in the prologue of the backward one, dW / db among finish_kernel's outputs.  C = 0: none.
struct HeadFwd {
  const float *W, *bias;   // (C,H), (C) or NULL
  float* logits;           // (B,C)
  int C, pad;
};
struct HeadBwd {
  const float *W, *dl;     // (C,H), dlogits (B,C)
  const float* hlast;      // final hidden state: row b at hlast + b * ldh
  float *dW, *db;          // (C,H), (C); may be NULL
  long long ldh;
  int C, pad;
};

struct FwdArgs {
  const float *gx, *VE, *UR, *EH, *h0, *c0;
  float *y, *hT, *cT, *gates, *cs, *Qs;
  float* trash;  // >= 64 floats: target of the redirected stores of inactive lanes (keeps stores unconditional)
  float* qxw;    // x-projection wave: qx rows for the weight-gradient kernels (written by the storer wave)
  int xwave;     // 1: wave NW computes the x-projection itself (XwArgs), no gx buffer is read
  unsigned* prog;  // training: the rows' progress words of the backward launch (WRide), cleared here
};
// second argument block of rec_fwd_kernel: what its x-projection wave needs (only that wave reads it, straight
// from the kernel-argument segment, so it costs the recurrent waves no registers)
// Cross-entropy of the riding classifier's logits (nn.CrossEntropyLoss of the reference's loop, train.py:58-65) in the same
// epilogue: every workgroup owns one batch row, so lse, the row's loss term and d(loss)/d(logits) = (softmax - onehot) / N are
// row-local; the mean comes out of ONE integer atomic per row (count + fixed-point sum, ce_epilogue; no float atomics).  tgt = NULL: none.
struct CeFwd {
  const long long* tgt;   // (B) class indices
  long long ignore;       // rows with this target contribute nothing
  float *loss, *nvalid;   // 1, 1: mean over the counted rows; their number N
  float *lse, *dz;        // (B), (B,C): row statistics; gradient of the logits for d(loss) = 1
  unsigned long long* ticket;   // two 8-byte words, zero between launches (the last workgroup puts them back)
};
struct XwArgs {
  const float *x, *UXP, *WXD, *BBT;
  HeadFwd hd;   // read from the kernel-argument segment by the epilogue only
  CeFwd ce;     // the same
  // direct mode (vmlmf_direct.inc; direct != 0): UXP / WXD / BBT point at the reference's own u_x (I, rw), v_x (4H, rw), b_x (4H),
  // and the x-projection wave also needs b_h (4H) and dia_x (1, I)
  const float *BH, *DX;
  int direct, pad;
};
// operands of the weight-gradient products (vmlmf_atb.inc)
struct AtbArgs {
  const float *dpre, *x, *y, *h0, *qx, *dqx, *Qs, *dQs;
  float* P;
  int pad0, pad;
};
// weight-gradient workers riding on rec_bwd_kernel's launch (vmlmf_atb.inc): K workers per task, chunks of S rows (t,b),
// ntg workgroups per worker index; prog = one progress word per batch row (zero between launches)
constexpr int WR_PROG_STRIDE = 32;   // unsigned words between two rows' progress words: one 128-byte line each (written through
                                     // every other step by 64+ workgroups: words sharing a line serialise at the memory side)
// codes a kernel leaves in the host-visible status word when one of its bounded waits gives up (the results are NaN then); the
// next C-ABI call on the device returns VMLMF_E_PROTOCOL (vmlmf_api.hip: take_status)
constexpr unsigned VMLMF_ST_WRIDE = 1, VMLMF_ST_CLUSTER = 2, VMLMF_ST_WF_FWD = 3, VMLMF_ST_WF_BWD = 4, VMLMF_ST_P2P = 5;
__device__ __forceinline__ void vg_raise(unsigned* status, unsigned code) {
  if (status != nullptr) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct WRide {
  AtbArgs a;
  // dux (or NULL): the riding workers also contract their x-fold tile with v_x - the tile's share of d(u_x), [worker k][task][16 x 16]
  // floats - so that ONE launch behind the recurrence can finish every gradient (finish2_kernel); vx: the reference's v_x (4H, rw)
  float* dux;
  const float* vx;
  unsigned* prog;
  int K, S, ntg, tasks;
  int lag;        // segments a progress word trails the stores it covers (a word is published every other segment)
  int direct;     // (of the whole BwdArgs, kept in this struct's padding) 1: VE / UE / EH are the reference's own v_h, u_h, dia_h
                  // and the rows' compute waves build their register images themselves (vmlmf_direct.inc)
  unsigned spin;  // looks at the progress words before a worker gives up (WR_SPIN; vmlmf_tune("test_wride_spin") shortens it)
  unsigned* status;   // host-visible status word (vmlmf_api.hip): VMLMF_ST_WRIDE is stored there when a worker gives up
};
struct BwdArgs {
  const float *gates, *cs, *c0, *dy, *dhT, *dcT, *VR, *UE, *EH;
  const float* VE;   // un-rotated V_h image (the forward's): rec3_bwd_kernel builds its own rotation from it
  float *dpre, *dQs, *dh0, *dc0;
  float* trash;
  HeadBwd hd;   // read from the kernel-argument segment by the prologue only
  WRide wr;     // K = 0: no riding workers
};
struct WgxArgs {
  const float *dpre, *VRX, *UXO, *EXI;
  float *dx, *dqx;
};
struct WghArgs {
  const float *dpre, *x, *y, *h0, *qx, *dqx, *Qs, *dQs;
  float* wpart;
};

// buffers of the step-wise path (vmlmf_generic.hip)
struct GenericBuf {
  // forward
  const float *gx, *EH, *h0, *c0, *Ud, *Vd, *zeros;
  float *y, *hT, *cT, *gates, *cs, *Qs, *Qtmp, *P, *ccar;
  // backward
  const float *dy, *dhT, *dcT, *UdT, *VdT, *VxT, *UXP, *EXT;
  float *dpre, *dQs, *dHrec, *ehterm, *dcar, *dh0, *dc0, *dqx, *dx;
  // split-K scratch of the GEMMs: partial products and per-tile tickets (zero between launches)
  float* part;
  long long part_cap;
  int* ticket;
  int ticket_cap;
};
int generic_forward(const VGeo& g, const GenericBuf& w, hipStream_t s);
int generic_backward(const VGeo& g, const GenericBuf& w, hipStream_t s);
// the non-recurrent tail of generic_backward on its own: dqx = dpre V_x over all rows, then dx (row-block kernels on large layers)
int generic_dqx_dx(const VGeo& g, const GenericBuf& w, hipStream_t s);
// qx = x U_x over all rows of a large time-major layer
int generic_qx(const VGeo& g, const float* x, const float* UXP, float* qx, hipStream_t s);

constexpr int RBX_MAXL = 4;   // layers of a clustered stack (vmlmf_rbx.hip)
// row-block MFMA recurrent kernels (vmlmf_rb.hip)
struct RbIo {
  const float *gx, *EH, *h0, *c0, *img, *dy, *dhT, *dcT;
  float *y, *hT, *cT, *gates, *cs, *Qs, *dpre, *dQs, *dh0, *dc0;
  float* xq;        // cluster exchange tiles (S > 1)
  unsigned* flag;   // cluster epoch words + error word
  unsigned* status; // host-visible status word (or NULL)
  int flags_zeroed;   // forward: rb_pack_kernel of this call has zeroed the epoch words (no memset node)
  DropArgs drop;      // dropout of the layer's output (state == nullptr: none): forward writes drop.yd, backward masks dy
};
// false: no instantiation covers the layer with S splits.  rows = live batch rows per workgroup (16, 8 or 4; 0 = automatic)
// xf = 1: with the x-side images and exchange tiles of the stacked form (vmlmf_rbx.inc); false when the layer is not covered by it
bool rb_geometry(const VGeo& g, int S, RbGeo* out, int rows = 0, int xf = 0);

// ---- clustered layers stacked in ONE launch per direction (vmlmf_rbx.inc): the layer loop of the LM network (vmlmf_lm.py:437-439)
// for layers too large for one CU.  Every layer keeps its own clusters (RbGeo, all layers alike) and forms its x side itself - the x
// ranks qx = x U_x are NPX more tiles of its exchange, the expansion a third MFMA product - so layer l + 1 needs nothing of layer l
// but the rows y_l[t] (their dropped copy under dropout), which it takes from L2 a few steps behind: a member reads the 16 units of
// its own tiles, written by the SAME member index of the layer below, whose epoch word says when they are complete.  Backward: the
// same with dx_l[t] going down.  Needs L x (row blocks) x S workgroups co-resident (one per CU).
struct RbxLayerF {
  const float *x, *EH, *EXT, *BBT, *h0, *c0, *img;
  float *y, *hT, *cT, *gates, *cs, *Qs, *qx;
  float* xq;               // this layer's exchange tiles
  unsigned* flag;          // ... and epoch words (+ error word behind them)
  const unsigned* pflag;   // epoch words of the layer below (its y is this layer's x), or NULL: x is complete before the launch
  int pub, pad;            // pub: a layer above consumes y during the launch (write-through stores + a final epoch)
  DropArgs drop;           // dropout of this layer's output (state == nullptr: none); the layer above reads drop.yd then
};
struct RbxLayerB {
  const float *gates, *cs, *dy, *dhT, *dcT, *EH, *EXT, *img;
  float *dpre, *dQs, *dqx, *dx, *dh0, *dc0;
  float* xq;
  unsigned* flag;
  const unsigned* pflag;   // epoch words of the layer ABOVE (its dx is this layer's dy), or NULL
  int pub, pad;            // pub: the layer below consumes dx during the launch
  DropArgs drop;           // dy is the gradient of the dropped copy: multiplied by the regenerated factors
};
struct RbxFwdArgs {
  unsigned* status;
  int L, bpl;              // layers; workgroups per layer (a multiple of 8)
  RbxLayerF l[RBX_MAXL];   // launch position 0 = the bottom layer
};
struct RbxBwdArgs {
  unsigned* status;
  int L, bpl;
  RbxLayerB l[RBX_MAXL];   // launch position 0 = the TOP layer (the producer comes first in the grid)
};
struct RbxZeroArgs {
  float* dpre[RBX_MAXL];
  unsigned* flags[RBX_MAXL];
};
bool rbx_supported(const VGeo& g, const RbGeo& q);   // an instantiation exists
int launch_rbx_fwd(const VGeo& g, const RbGeo& q, const RbxFwdArgs& a, hipStream_t s);
int launch_rbx_bwd(const VGeo& g, const RbGeo& q, const RbxBwdArgs& a, hipStream_t s);
// dpre of the padded slots of every layer zeroed and the backward launch's epoch words cleared, one launch for the stack
int launch_rbx_zero(const VGeo& g, const RbGeo& q, int L, float* const* dpre, unsigned* const* flags, hipStream_t s);
int launch_rb_pack(const VGeo& g, const RbGeo& q, const RefP& p, float* img, hipStream_t s, unsigned* zero_flags = nullptr);
int launch_rb_fwd(const VGeo& g, const RbGeo& q, const RbIo& io, hipStream_t s);
// rb_pack_kernel for every layer of a clustered stack (all layers share the geometry) in one launch
int launch_rb_pack_stack(const VGeo& g, const RbGeo& q, int L, const RefP* p, float* const* img, unsigned* const* zero_flags, hipStream_t s);
int launch_rb_bwd(const VGeo& g, const RbGeo& q, const RbIo& io, hipStream_t s);

// every launcher returns hipGetLastError() of its launch, or VMLMF_E_UNSUPPORTED (-3) when no
// instantiation covers the geometry
int launch_pack(const VGeo& g, const RefP& p, const VPack& L, float* pack, hipStream_t s);
int launch_xproj(const VGeo& g, const VPack& L, const float* pack, const float* x, float* gx, float* qx,
                 hipStream_t s);
int launch_rec_fwd(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s);
int launch_rec_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s);
// third form of the recurrent kernels (vmlmf_rec3.inc): one-group layers, padded hidden rank <= 16, <= 3 waves of units; the
// forward also needs the x-projection wave's envelope (narrow input, x-fold).  Same tapes as the kernels above.
// rec_bwd_kernel has a form with the riding weight-gradient workers for this layer (wider ranks / four waves of units: what
// rec3_bwd_kernel does not cover)
inline bool rec_bwd_rides(const VGeo& g) { return g.KH > 16 || g.W > 3; }
bool rec3_fwd_supported(const VGeo& g);
bool rec3_bwd_supported(const VGeo& g);
int launch_rec3_fwd(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s);
int launch_rec3_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s);
// fourth form of the backward (vmlmf_rec4.inc): rec3_bwd_kernel's layers with the x-fold; the weight gradients are formed inside
// the rows' workgroups (a.wr.a: operands and the partial blocks, one per workgroup; a.wr.K = 0), no dpre / dQ is written
bool rec4_bwd_supported(const VGeo& g);
int launch_rec4_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s);
int launch_wgrad_x(const VGeo& g, const WgxArgs& a, hipStream_t s);
int launch_wgrad_h(const VGeo& g, const WghArgs& a, hipStream_t s);
bool wgrad_ring_ok(const VGeo& g);   // vmlmf_wgrad_ring.hip: the same products for large layers, operands through an LDS ring
int launch_wgrad_ring(const VGeo& g, const WghArgs& a, int cus, int nc_out[3], hipStream_t s);
// partial blocks that hold the C1 (+ E) / C2 / C3 region of the chunk layout (wgrad_ring_kernel chunks each product on its own);
// c[0] == 0: g.nchunk blocks hold every region.  (A kernel argument of the reduction alone: VGeo is every kernel's first argument.)
struct ReduceCounts {
  int c[3];
};
int launch_reduce(const VGeo& g, const float* wpart, float* cgrad, unsigned* prog, hipStream_t s,
                  ReduceCounts wc = ReduceCounts{{0, 0, 0}});   // prog: words to clear, or NULL
// reduce + finish of a launch with riding workers in ONE launch (vmlmf_pack.hip: finish2_kernel): K partial blocks and the workers'
// d(u_x) shares -> the reference-layout gradients; prog: the rows' progress words, cleared here
bool finish2_ok(const VGeo& g);
int launch_finish2(const VGeo& g, const RefP& p, const float* wpart, const float* dux, int K, const RefG& out, const HeadBwd& hd,
                   unsigned* prog, hipStream_t s, unsigned* health);
int launch_finish(const VGeo& g, const RefP& p, const float* cgrad, const RefG& out, const HeadBwd& hd, hipStream_t s,
                  unsigned* health = nullptr);   // health: device word set when a gradient written is not finite (or NULL)
// the library's per-device gradient-health word (vmlmf_api.hip), what the optimizers' step guard reads; NULL before the first
// training call on the device
unsigned* vmlmf_health_word_if_any();
int vmlmf_adam_guard_mode();   // 1: health word (default), 2: a scan launch over the gradients, 0: none

// classifier head (vmlmf_head.hip)
int head_max_classes();
hipError_t launch_head_fwd(int B, int H, int C, const float* h, long long ldh, const float* W, const float* bias,
                           float* out, hipStream_t s);
hipError_t launch_head_bwd(int B, int H, int C, const float* h, long long ldh, const float* W, const float* dl,
                           float* dh, float* dW, float* db, hipStream_t s);

hipError_t launch_ce_fwd(int B, int C, const float* z, const long long* tgt, long long ignore_index, float* loss,
                         float* lse, float* nvalid, float* dz_unit, hipStream_t s);
hipError_t launch_ce_bwd(int B, int C, const float* z, const long long* tgt, long long ignore_index,
                         const float* lse, const float* nvalid, const float* dloss, float* dz, hipStream_t s);

// softmax negative log-likelihood over the vocabulary (vmlmf_nll.hip)
hipError_t launch_nll_fwd(int R, int V, const float* scores, const long long* y, float scale, float* loss, float* lse,
                          float* rowloss, hipStream_t s);
hipError_t launch_nll_bwd(int R, int V, const float* scores, const long long* y, float scale, const float* lse,
                          const float* dloss, float* dscores, hipStream_t s);

// training form of the loss: loss, the scores' gradient (in place, for d(loss) = 1) and the bias gradient in one pass
int nll_grad_workgroups(int R);
int launch_nll_fwd_grad(int R, int V, float* scores, const float* bias, const long long* y, float scale, float* loss,
                        float* rowloss, float* dbias, float* scratch, hipStream_t s);
// embedding-table gradient (vmlmf_embed.hip)
size_t embed_bwd_scratch_bytes(int R, int V);
int launch_transpose(int rows, int cols, const float* src, float* dst, hipStream_t s);   // dst (cols x rows) = src^T
int launch_embed_bwd(int R, int H, int V, const long long* tokens, const float* dy, float* dW, void* scratch, size_t scratch_bytes,
                     hipStream_t s, const DropArgs* drop = nullptr);
// dropout launches (vmlmf_dropout.hip): snapshot + advance of the generator state; mode 0: y = x * factor, 1: y = factor, 2: y =
// w[tokens] * factor over R positions of H columns
int launch_drop_advance(unsigned long long* state, unsigned long long* snap, hipStream_t s);
int launch_drop_rows(int mode, long long R, int H, int V, const DropArgs& d, const DropCols& cm, const float* x, const long long* tokens, float* y,
                     hipStream_t s);

// ---- wavefront kernels for stacked layers (vmlmf_wave.inc) ----
// One launch runs every layer of a stack: workgroup = (layer, batch row).  Besides the recurrence's compute waves a
// workgroup has an "x-team" of as many waves that forms the layer's x-side pre-activations in-kernel (forward) / the
// gradient of the layer's input (backward), so a layer above the first consumes the rows of the layer below as that one
// produces them (progress words in L2), a few steps behind.
constexpr int WF_MAXL = 4;          // layers per launch
constexpr int WF_FLAG_STRIDE = 32;  // unsigned words between two progress words (one 128-byte line each)
struct WfPack {                     // float offsets inside the WF region of PACK (images with the half-pass layout)
  long long UR, VR, URX, VRX;
  long long VE, UE, VXK, UXK;       // only when padded w_rank != padded u_rank: pack_kernel's VE / UE / VXT / UXO re-laid to
                                    // the common width K = max of the two (zero ranks behind the narrower side)
  long long total;
};
__host__ __device__ inline int wf_width(const VGeo& g) { return g.KH > g.KX ? g.KH : g.KX; }   // rank width the wavefront kernels are instantiated for
struct WfFwdLayer {
  const float* x;                   // input rows of the layer (above the first: the y of the layer below)
  const float *VE, *EH, *VXT, *EXT, *BBT;   // pack_kernel images
  const float *UR, *URX;            // wf_pack_kernel images
  const float *h0, *c0;
  float *y, *hT, *cT, *gates, *cs, *Qs, *qx;
  long long sxT, sxB;
  long long syT, syB;               // strides of this layer's y (round 6: layers of a stack may differ in hidden_size)
  int I, H;                         // input_size, hidden_size of THIS layer
  int Hg, pad;                      // hidden units per group
};
struct WfBwdLayer {
  const float *gates, *cs, *dy, *dhT, *dcT;
  const float *UE, *EH, *UXO, *EXI; // pack_kernel images
  const float *VR, *VRX;            // wf_pack_kernel images
  float *dpre, *dQs, *dqx, *dx, *dh0, *dc0;
  long long sxT, sxB;               // strides of dx (= the layer's x)
  long long syT, syB;               // strides of dy (= the layer's y)
  int I, want_dx;
  int H, Hg;                        // hidden_size, units per group of THIS layer
};
struct WfCommon {
  unsigned* flag;                   // [L - 1][B][WF_FLAG_STRIDE] progress words, then the error word
  unsigned* status;                 // host-visible status word (or NULL)
  int L, pad;
};
// drop[pos]: nn.Dropout behind the layer at launch position pos (vmlmf_lm.py:438-439; state == nullptr: none), round 6.  Forward: the
// layer's loader wave forms the factors of a step (one Philox call covers the row's NT columns: lane l the quad l), its storer writes
// y AND the dropped copy drop.yd (what the layer above / the caller reads).  Backward: the loader forms the same factors beside the
// tape, the compute waves multiply the incoming dy.  Kept apart from the layer blocks: only those two roles read it.
struct WfFwdArgs {
  WfCommon c;
  WfFwdLayer l[WF_MAXL];
  HeadFwd hd;   // classifier on the top layer's final hidden state (C = 0: none)
  DropArgs drop[WF_MAXL];
};
struct WfBwdArgs {
  WfCommon c;
  WfBwdLayer l[WF_MAXL];
  HeadBwd hd;
  DropArgs drop[WF_MAXL];
};
bool wf_supported(const VGeo& g);   // an instantiation exists for the layer's (rank, waves)
WfPack wf_pack_layout(const VGeo& g);
// every layer's parameter images (pack_kernel's + the rotated ones at PACK + VPack::WF) in one launch; zero0 / zero1:
// progress words to clear (the forward's and the backward's), or NULL
int launch_pack_stack(int L, const VGeo* g, const RefP* p, const VPack* P, const WfPack& W, float* const* pack, unsigned* zero0,
                      int nzero0, unsigned* zero1, int nzero1, hipStream_t s, int images = 0);
// `images`: which of pack_kernel's images the launch produces (the dot elements EH / EXI / EXT and the rotated WF region always)
enum { PACK_ALL = 0, PACK_CLUSTERED = 1, PACK_WAVEFRONT = 2 };   // slim: EH / EXT / BBT only
int launch_wf_fwd(const VGeo& g, const WfFwdArgs& a, hipStream_t s);
int launch_wf_bwd(const VGeo& g, const WfBwdArgs& a, hipStream_t s);
// the batched half of the backward of every layer of a stack, one launch each (grid.z / grid.y = layer)
int launch_wgrad_h_stack(int L, const VGeo* g, const WghArgs* w, hipStream_t s);
struct AtbStack;   // (vmlmf_atb.inc) the layers' geometry + argument blocks of a weight-gradient launch over a stack
// vmlmf_wgrad4.hip: four interleaved column tiles per wave; -3 when the stack is outside its envelope (the caller launches wgrad_mfma_stack_kernel)
int launch_wgrad4_stack(int L, const VGeo* g, const WghArgs* w, const AtbStack& S, hipStream_t s);
int wgrad4_chunk_rows(int L, const VGeo* g, int cus);   // rows per chunk for a stack that kernel will take; 0: not its stack
int launch_reduce_stack(int L, const VGeo* g, const float* const* wpart, float* const* cgrad, hipStream_t s,
                        const ReduceCounts* wc = nullptr);   // wc: per layer, the chunk counts of a wgrad_ring_kernel launch (or NULL)
// wpart (+ wc, as launch_reduce_stack): finish straight from the partial blocks - no reduce launch in front (finish_from_blocks_ok layers)
bool finish_from_blocks_ok(const VGeo& g);
int launch_finish_stack(int L, const VGeo* g, const RefP* p, const float* const* cgrad, const RefG* out, const HeadBwd& hd_top,
                        hipStream_t s, unsigned* health = nullptr, const float* const* wpart = nullptr,
                        const ReduceCounts* wc = nullptr);
// reduce_cg_stack_kernel + finish_stack_kernel as one launch (a workgroup per hidden unit: vmlmf_pack.hip); -3: not for these layers
bool finish_units_ok(const VGeo& g);
int launch_finish_units_stack(int L, const VGeo* g, const RefP* p, const RefG* out, const HeadBwd& hd_top, hipStream_t s,
                              unsigned* health, const float* const* wpart, const ReduceCounts* wc);   // hd_top: classifier gradients ride with the top layer (C = 0: none)

// register budget of the persistent kernels: which (KH, NT) pairs are instantiated
bool rec_supported(const VGeo& g);
