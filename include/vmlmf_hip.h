/*
 * vmlmf_hip.h - C ABI of the MI355X (gfx950) VMLMF compressed-LSTM hot path.
 *
 * The reference (snudm-starlab/VMLMF) has no FFI: its hot path is Python over ATen.  The functions below
 * are what a binding for that path would call; each one names the reference code it replaces
 * (V/ = rnn_compression_factorization_vmlmf/).  Everything is plain pointers + sizes: no torch types.
 * All pointers are DEVICE pointers (fp32) unless stated; all work is enqueued on `stream` (a hipStream_t
 * passed as void*), nothing synchronises.  Return value: 0 = ok, <0 = VMLMF_E_*, >0 = hipError_t.
 */
#ifndef VMLMF_HIP_H
#define VMLMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMLMF_ABI_VERSION 13
#define VMLMF_MAX_G 2

/* cell variants */
#define VMLMF_V1_CELL 1       /* MyVMLMFCell       V/src/models/vmlmf.py:38-125        */
#define VMLMF_V2_GROUP_CELL 2 /* MyVMLMFCellg2     V/src/models/vmlmf_group.py:37-155  */
#define VMLMF_V3_LM 3         /* MyVMLSTM          V/src/models/vmlmf_lm.py:178-280    */
#define VMLMF_V4_LM_GROUP 4   /* MyVMLSTMGroup     V/src/models/vmlmf_lm.py:53-174     */
/* the reference's comparison cells without the vector multiplication (no dia_*, no diagonal removal) */
#define VMLMF_V5_LMF_CELL 5   /* MyLSTMCell, low-rank mode   V/src/models/vmlmf.py:159-186,198-224   */
#define VMLMF_V6_GROUP_NOVM 6 /* MyVMLMFgCellg2 (ablation)   V/src/models/vmlmf_group.py:158-251     */

#define VMLMF_E_BADARG (-1)      /* null pointer / inconsistent descriptor                          */
#define VMLMF_E_SHAPE (-2)       /* shape the reference itself rejects (I > H, I != H for LM, H % g) */
#define VMLMF_E_UNSUPPORTED (-3) /* valid for the reference, not yet covered by the HIP kernels       */
#define VMLMF_E_WORKSPACE (-4)   /* workspace / reserve smaller than vmlmf_query() asked for          */
#define VMLMF_E_COMM (-5)        /* RCCL reported an error (text in vmlmf_last_error())                */
#define VMLMF_E_PROTOCOL (-6)    /* ABI 8: a launch gave up a bounded wait for another workgroup (riding weight-gradient workers,
                                  * row-block clusters, wavefront hand-overs).  Its results are NaN (never a plausible wrong
                                  * number); nothing in the launch hangs.  Launches are asynchronous, so the code comes back from
                                  * the NEXT forward / backward / stack call on the device (from the failing call itself under
                                  * VMLMF_DEBUG_SYNC=1), or from vmlmf_check_status() once the stream has been synchronised.   */

/* One layer's problem description.  x is (T,B,I) when time_major else (B,T,I); y likewise with H. */
typedef struct vmlmf_desc {
  int32_t variant;           /* VMLMF_V*                                                        */
  int32_t B, T, I, H;        /* batch, time steps, input_size, hidden_size                      */
  int32_t w_rank;            /* rank of the input->hidden factorisation  (u_x: I x w_rank)      */
  int32_t g;                 /* groups of the hidden->hidden path (1 for V1/V3/V5, 2 for V2/V4/V6) */
  int32_t u_ranks[VMLMF_MAX_G]; /* rank per shift s (V1/V3/V5: only [0])                        */
  int32_t time_major;        /* 1: (T,B,*)  LM layers;  0: (B,T,*)  MyLSTM batch_first          */
  int32_t training;          /* 1: forward fills `reserve` for backward; 0: inference            */
  int32_t dtype;             /* VMLMF_DT_F32 (0): the reference's arithmetic.  VMLMF_DT_BF16 (1), ABI 4: both products of a
                              * step on bf16 MFMA (weights and the activations entering an MFMA rounded to bf16, fp32
                              * accumulate), bf16 tapes for the x-side pre-activations, the gates and dpre; c, every sum
                              * and all weight gradients fp32; x, y, states, parameters and gradients stay fp32 tensors.
                              * Implemented by the row-block kernels for one-group layers (V1, V3, V5) within one CU's
                              * registers; VMLMF_E_UNSUPPORTED elsewhere.  Tolerance: tests/test_gpu_bf16.py              */
} vmlmf_desc;
#define VMLMF_DT_F32 0
#define VMLMF_DT_BF16 1

/* Parameters in the REFERENCE's own layouts (so a state_dict's tensors are passed as they are):
 *   V1: u_x (I,rw) v_x (4H,rw) u_h[0] (H,ru) v_h[0] (4H,ru) b_x b_h (4H) dia_x (1,I) dia_h (1,H)     vmlmf.py:56-69
 *   V2: u_x v_x as V1; u_h[s] (g,H/g,ru_s) v_h[s] (g,ru_s,4H/g); b_x=bias_x b_h=bias_h (1,4H)        vmlmf_group.py:61-79
 *   V3: as V1 with v_x=w_x, v_h[0]=w_h                                                               vmlmf_lm.py:200-213
 *   V4: as V2 with v_x=w_x, b_x/b_h (4H)                                                             vmlmf_lm.py:77-91
 *   V5: u_x=w (I,rw) u_h[0]=u (H,ru); per gate k in (i,f,o,c~) order: w_gate[k]=w{k+1} (rw,H),
 *       u_gate[k]=u{k+1} (ru,H), b_gate[k]=bias_i/bias_f/bias_o/bias_c (1,H); everything else NULL        vmlmf.py:159-186
 *   V6: as V2 without dia_x/dia_h (NULL); its x side chunks (f,i,n,o) like its h side                 vmlmf_group.py:183-197,211
 * The same struct (non-const view) receives the gradients in the same layouts. */
typedef struct vmlmf_params {
  const float *dia_x, *dia_h, *u_x, *v_x, *b_x, *b_h;
  const float *u_h[VMLMF_MAX_G];
  const float *v_h[VMLMF_MAX_G];
  const float *w_gate[4], *u_gate[4], *b_gate[4]; /* V5 only (ABI 2) */
} vmlmf_params;

typedef struct vmlmf_grads {
  float *dia_x, *dia_h, *u_x, *v_x, *b_x, *b_h;
  float *u_h[VMLMF_MAX_G];
  float *v_h[VMLMF_MAX_G];
  float *w_gate[4], *u_gate[4], *b_gate[4];       /* V5 only (ABI 2) */
} vmlmf_grads;

typedef struct vmlmf_sizes {
  size_t workspace_bytes; /* scratch, may be reused by the next call on the same stream            */
  size_t reserve_bytes;   /* written by forward(training=1), read by backward                      */
  int32_t rows_per_wg;    /* batch rows one persistent workgroup owns                              */
  int32_t threads_per_wg; /* = groups * waves_per_group * 64                                       */
  int32_t workgroups;     /* grid of the two recurrent kernels                                     */
  int32_t kx, kh;         /* padded rank widths held in registers                                  */
} vmlmf_sizes;

/* ABI / build identification. */
int vmlmf_abi_version(void);
const char *vmlmf_build_info(void);
/* Text of the last error raised on this host thread (never NULL). */
const char *vmlmf_last_error(void);

/*
 * Kernel-selection switches (A/B measurements and tests; process-wide, take effect at the next call, and a descriptor
 * must see the same setting in vmlmf_query, forward and backward).  Keys:
 *   "rb"            -1 automatic (default), 0 never, 1 always where instantiated: the row-block recurrent kernels
 *                   (16 batch rows per workgroup, both products of a step on v_mfma_f32_16x16x4_f32) instead of the
 *                   one-row-per-CU VALU kernels / the step-wise path
 *   "rb_min_batch"  batch size from which automatic mode picks them for layers the VALU kernels also cover (default 0 =
 *                   never: measured, the VALU kernels win there at every batch size; automatic mode uses the row-block
 *                   kernels only for layers beyond one CU's registers, e.g. H = 650)
 *   "rb_cluster"    workgroups a 16-row block's hidden units are split over for layers beyond one CU (0 = automatic)
 *   "rb_rows"       live batch rows of a row-block workgroup: 16, 8 or 4 of the 16 MFMA columns (0 = automatic); fewer rows
 *                   = more workgroups, each streaming fewer tape bytes through its CU
 *   "rec3"          bit mask of the round-3 recurrent kernels (default 6): 1 forward always, 2 backward, 4 forward when the
 *                   batch has more rows than the device CUs
 *   "wride"         1 (default): the weight-gradient products ride on the backward recurrence's launch where they fit;
 *                   0: always the stand-alone kernel behind it.  The library sets 0 by itself after a riding worker gave up
 *                   its bounded wait (VMLMF_E_PROTOCOL; a GPU shared with other processes can starve the workers of their
 *                   rows); 1 re-arms
 *   "inrow"         the backward that forms the weight gradients inside the rows' workgroups (no dpre tape, no weight-gradient
 *                   launch; layers with the x-fold whose input needs no gradient): -1 (default) automatic - batches beyond the
 *                   riding workers' range -, 0 never, 1 wherever it covers the layer
 *   "adam_guard"    how vmlmf_adam_step_guarded finds non-finite gradients: 1 (default) the health word finish_kernel sets,
 *                   2 a scan launch over the listed gradients, 0 not at all
 *   "clear_health"  (any value) clear the gradient-health word: it stays set from a backward that wrote non-finite gradients until a
 *                   guarded optimizer step consumes it - a caller that handled such a step some other way says so here, or the
 *                   next guarded step is skipped once
 *   "direct"        1 (default): layers of the V1 / V3 layouts with narrow inputs and ranks 8 / 16 build their register images inside
 *                   the recurrent kernels' prologues from the reference layouts (no pack_kernel launch in the call); 0: always pack
 *   "finish2"       1 (default): behind a backward whose weight-gradient workers rode on the recurrent launch, ONE launch sums their
 *                   partial blocks and writes the reference-layout gradients (finish2_kernel); 0: reduce_cg_kernel + finish_kernel
 *   "wring"         the batched weight-gradient products of large layers (thread slots >= 256, fp32 tapes, time-major contiguous x / y)
 *                   with their operands streamed through an LDS ring (wgrad_ring_kernel): -1 (default) for the layers of the
 *                   step-wise / clustered recurrences with >= 1024 rows, 0 never, 1 wherever the kernel takes the layer
 *   "rbx"           1 (default): vmlmf_stack_* runs two to four clustered layers (hidden units beyond one CU, e.g. the PTB layers) in one
 *                   launch per direction while every layer's clusters are co-resident; 0: never (the caller chains the layers);
 *                   2: a single such layer takes that form too (measurements).  VMLMF_RBX in the environment
 *   "ffb"           0 (default): behind the clustered stack's backward, reduce_cg_stack_kernel + finish_stack_kernel; 1: one finishing
 *                   launch that sums the partial blocks itself (measured slower; kept parity-tested).  VMLMF_FFB
 *   "test_wride_spin"  looks a riding worker takes before it gives up (tests of the failure path; 0 = the production bound)
 */
int vmlmf_tune(const char *key, int value);

/* ABI 8.  0, or VMLMF_E_PROTOCOL when a launch that has already run on the current device gave up a bounded wait (text in
 * vmlmf_last_error(); the condition is cleared).  Host only: reads a status word in mapped host memory, no GPU call. */
int vmlmf_check_status(void);

/* Validate `d` and report buffer sizes + launch geometry.  Host only, no GPU call. */
int vmlmf_query(const vmlmf_desc *d, vmlmf_sizes *out);

/*
 * Sequence forward of one layer: replaces the Python time loop + cell
 *   MyLSTM.forward             V/src/models/vmlmf.py:300-314   (h0 = c0 = NULL -> zeros, 302-303)
 *   MyVMLSTM[Group].forward    V/src/models/vmlmf_lm.py:272-280, 166-174  (states passed in)
 * T = 1 is the bare cell call  MyVMLMFCell.forward vmlmf.py:78-125 / lstm_step vmlmf_lm.py:222-269.
 * y: all hidden states; hT,cT: (B,H) final state (may be NULL).  reserve may be NULL iff !training.
 */
int vmlmf_seq_forward(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                      const float *c0, float *y, float *hT, float *cT, void *reserve, void *workspace,
                      size_t workspace_bytes, void *stream);

/*
 * Kept parameter images (ABI 5).  vmlmf_seq_forward turns the reference-layout parameters into the kernels' register
 * images on every call (pack_kernel: 6 us of the 185 us headline step).  A caller whose parameters have not changed since an
 * earlier call - inference, evaluation, gradient accumulation, a benchmark loop without an optimizer - can keep the images:
 *   vmlmf_pack_bytes(desc, &n); vmlmf_pack_params(desc, params, packed, stream);
 *   vmlmf_seq_forward_packed(..., packed) / vmlmf_seq_backward_packed(..., packed)      (packed == NULL: the plain calls)
 * `packed` must have been made for the same descriptor (batch and sequence length may differ) and must stay unchanged until the
 * backward that uses it has run; it is the caller's duty to re-pack after ANY change of the parameters.  Not offered for
 * the step-wise / clustered layers (VMLMF_E_UNSUPPORTED): their image carries per-call state.  vmlmf_tune_generation()
 * counts vmlmf_tune() calls (and the library's own switch after a failed riding launch): images made under an older
 * generation may have another layout.  ABI 9: the library keeps, on the host, what every image was packed for (its device
 * address -> variant, sizes, ranks, kernel family, generation; the last 256 images of the process); a *_packed call with an
 * address vmlmf_pack_params did not fill, or with an image packed for another descriptor (batch and sequence length apart)
 * or under another generation, returns VMLMF_E_BADARG and launches nothing.  What it cannot see is a change of the PARAMETER
 * VALUES since the image was packed: re-packing after an optimizer step stays the caller's duty.
 */
int vmlmf_pack_bytes(const vmlmf_desc *d, size_t *bytes);
int vmlmf_pack_params(const vmlmf_desc *d, const vmlmf_params *p, void *packed, void *stream);
int vmlmf_seq_forward_packed(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                             const float *c0, float *y, float *hT, float *cT, void *reserve, void *workspace,
                             size_t workspace_bytes, void *stream, const void *packed);
int vmlmf_seq_backward_packed(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                              const float *c0, const float *y, const void *reserve, const float *dy,
                              const float *dhT, const float *dcT, float *dx, float *dh0, float *dc0,
                              const vmlmf_grads *g, void *workspace, size_t workspace_bytes, void *stream,
                              const void *packed);
int vmlmf_tune_generation(void);
/* ABI 11: the current value of a vmlmf_tune switch.  "wride" reads 0 while the riding weight-gradient workers are off - by
 * VMLMF_WRIDE=0, by vmlmf_tune("wride", 0), or because a launch gave up a bounded wait (VMLMF_E_PROTOCOL) and the library fell back
 * to the stand-alone kernel: a benchmark reports it, so a shared GPU cannot pass for a regression. */
int vmlmf_tune_get(const char *key, int *value);

/*
 * Classifier riding on a layer (ABI 6): Net applies nn.Linear(H, 18) to the last layer's final hidden state
 * (V/src/models/vmlmf.py:345,353-355).  Given with the layer's forward / backward it costs no launch of its own: the
 * logits come out of the epilogue of the forward recurrence, d(hT) = dlogits W is formed in the prologue of the backward
 * one, dweight / dbias are extra outputs of the final gradient kernel (two launches and their boundaries less per training
 * step: 0.1787 -> 0.167 ms at the headline shape).  Layers on the row-block / step-wise kernels get the same results from
 * the stand-alone head kernels, launched inside the call.  classes <= 32.
 *   forward : weight (C,H), bias (C) or NULL, logits (B,C) out
 *   backward: weight, dlogits (B,C) in; dweight (C,H), dbias (C) out (either may be NULL).  The gradient that reaches
 *             the final hidden state is dhT (if given) + dlogits W.
 * vmlmf_extra gathers the optional arguments of a call; NULL members are "not used".
 */
typedef struct vmlmf_head {
  int32_t classes;
  const float *weight, *bias;
  float *logits;
  const float *dlogits;
  float *dweight, *dbias;
} vmlmf_head;
/* ABI 10: the criterion of the reference's loop (nn.CrossEntropyLoss() with default arguments on Net's output, V/src/train_test/
 * train.py:58-65) riding on the same forward launch as the classifier: the logits of a batch row never leave the workgroup that
 * formed them before the row's log-sum-exp, loss term and d(loss)/d(logits) = (softmax - onehot) / N exist; the mean over the N
 * rows whose target is not ignore_index is an integer sum of the rows' terms in fixed point (2^-29 at 64 rows: associative, so
 * run-to-run identical whatever order the rows finish in, with one atomic per row and no pass over the rows; terms of 2048 and
 * more are summed at 2^-10, beyond 2^36 / B the loss is +Inf, a NaN term makes it NaN).  Forward only, together with `head`; values as vmlmf_ce_forward's on the same logits (the mean's summation order
 * differs).  On the layer families whose classifier is a launch of its own (row-block, step-wise) the criterion is one too. */
typedef struct vmlmf_ce {
  const int64_t *target;     /* (B) class indices; an index outside [0, classes) poisons the loss with NaN               */
  int64_t ignore_index;
  float *loss, *nvalid;      /* 1, 1                                                                                       */
  float *lse;                /* (B)                                                                                        */
  float *dlogits_unit;       /* (B, classes) gradient of the logits for d(loss) = 1, or NULL                               */
  uint64_t *ticket;          /* TWO 8-byte words that are zero before the first launch; every launch leaves them zero.
                              * Launches that share them must be ordered on one stream                                     */
} vmlmf_ce;
/* ABI 11: the dropout behind a layer of the LM network (`x = self.dropout(x)` of V/src/models/vmlmf_lm.py:438-439, nn.Dropout(p)
 * of :402) inside the layer's own launches: the forward writes y (kept for the backward and the carried state) AND its dropped copy
 * y_dropped = y * factor, factor = 0 with probability p and 1/(1-p) otherwise; the backward takes dy as the gradient of y_dropped and
 * multiplies it by the same factors, regenerated from (state, site) - no mask tensor, no launch.  The factors are a pure function of
 * (state[0] = seed, state[1] = offset, site, position t*B + b, column): Philox4x32-10, csrc/vmlmf_dropout.h; vmlmf_dropout_factors()
 * returns them as a tensor (tests: the oracle multiplies by it).  Only layers for which vmlmf_dropout_fused() is 1; others:
 * VMLMF_E_UNSUPPORTED, use vmlmf_dropout_apply behind the layer. */
typedef struct vmlmf_dropout {
  float p;                   /* drop probability, [0, 1)                                                                     */
  int32_t site;              /* which dropout of the network (a different stream of factors per site)                       */
  const int64_t *state;      /* device: {seed, offset} as vmlmf_dropout_advance snapshotted them for this forward           */
  float *y_dropped;          /* forward: layout of y; backward: unused                                                       */
} vmlmf_dropout;
typedef struct vmlmf_extra {
  const void *packed;        /* kept parameter images (vmlmf_pack_params) or NULL */
  const vmlmf_head *head;    /* classifier on the final hidden state or NULL      */
  const vmlmf_ce *ce;        /* ABI 10: cross-entropy on that classifier's logits (forward calls; needs `head`) or NULL */
  const vmlmf_dropout *drop; /* ABI 11: dropout of the layer's output inside its launches, or NULL                     */
} vmlmf_extra;
int vmlmf_seq_forward_ex(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                         const float *c0, float *y, float *hT, float *cT, void *reserve, void *workspace,
                         size_t workspace_bytes, void *stream, const vmlmf_extra *ex);
int vmlmf_seq_backward_ex(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                          const float *c0, const float *y, const void *reserve, const float *dy,
                          const float *dhT, const float *dcT, float *dx, float *dh0, float *dc0,
                          const vmlmf_grads *g, void *workspace, size_t workspace_bytes, void *stream,
                          const vmlmf_extra *ex);

/*
 * Sequence backward: replaces autograd's replay of the ~75 ATen ops per timestep.
 * dy: upstream gradient of y (same layout as y, may be NULL = zeros); dhT,dcT (B,H) may be NULL.
 * Outputs: dx (layout of x, may be NULL), dh0,dc0 (B,H, may be NULL), and every parameter gradient in
 * `g` (overwritten, reference layouts; deterministic summation order).  `reserve` and `y` must be the
 * ones the matching forward produced; `workspace` may be a different buffer.
 */
/* ---- stacked layers in one launch per direction (ABI 7) -------------------------------------------------------
 * The layer loop of MyLSTM.forward (V/src/models/vmlmf.py:300-314: `x = h` between the cells of a time step) and of the
 * LM network (V/src/models/vmlmf_lm.py:437-439) as a WAVEFRONT: every layer's recurrence runs in the same launch, layer
 * l+1 consuming the hidden states of layer l a few time steps behind it, and each layer forms its x-side products inside
 * that launch (no x-projection / input-gradient launches, no (T,B,4H) pre-activation round trip).  Results are those of
 * L calls of vmlmf_seq_forward / vmlmf_seq_backward chained through y (same arithmetic per element; the order of the
 * rank-space sums differs in the last bits).
 * Covered: 1..4 layers of V1, V2, V3, V5 or V6 (not the flat V4 layout) with equal B, T, H, ranks and layout, at most four
 * waves of hidden units (hidden_size <= 256; two groups: hidden_size / 2 <= 128), the wider of padded w_rank and (summed)
 * padded u_rank <= 24, or <= 32 with at most three waves of units, layer l > 0 with input_size == the hidden_size of layer l - 1.
 * Round 6: the layers may differ in hidden_size (MyLSTM builds any hidden_layer_sizes, vmlmf.py:283-292) - one-group layers with one
 * padded rank on both sides and fp32 tapes; every layer then runs on the widest layer's thread-slot geometry.  Anything else:
 * VMLMF_E_UNSUPPORTED from vmlmf_stack_query() - the caller then chains the per-layer calls.
 * Per layer: desc (training flag and shapes must agree across the stack), params, optional initial / final states, the
 * layer's output y (B,T,H or T,B,H; layer l's y is layer l+1's x) and its reserve (training).  Backward additionally:
 * gradients of the final states (or NULL), of the initial states (or NULL) and the parameter gradients. */
#define VMLMF_STACK_MAX 4
typedef struct vmlmf_stack_layer {
  vmlmf_desc desc;
  const vmlmf_params *params;
  const float *h0, *c0;
  float *y, *hT, *cT;
  void *reserve;
  const float *dhT, *dcT;
  float *dh0, *dc0;
  const vmlmf_grads *grads;
  const vmlmf_dropout *drop; /* ABI 12: dropout of this layer's output inside the stack's launches (vmlmf_lm.py:438-439), or NULL:
                              * the layer above (and the caller, for the top layer) reads drop->y_dropped instead of y.  The clustered
                              * form and the wavefront stacks of one-group layers take it (vmlmf_stack_dropout_fused) */
} vmlmf_stack_layer;
/* ABI 12 - a second form behind the same three entry points: layers too large for one CU (the PTB layers, hidden_size 650) run on
 * CLUSTERS of workgroups (vmlmf_seq_forward's row-block kernels); stacked, every layer keeps its own clusters inside ONE launch per
 * direction and layer l + 1 follows layer l a few time steps behind (csrc/vmlmf_rbx.hip) - T + 3 dependent cluster steps per
 * direction instead of L x T - and each layer forms its x side inside that launch as well.  Covered: 2..4 V3 or V4 layers of one
 * configuration, fp32, time-major, input_size == hidden_size for every layer, w_rank 17..32, as long as the clusters of all layers
 * are co-resident (L x ceil(B / rows) x 16 workgroups <= the device's CUs: up to 128 rows for two layers).  vmlmf_stack_query()
 * returns VMLMF_E_UNSUPPORTED otherwise.  No classifier head on this form. */
/* 1: the layers' `drop` fields are honoured by this stack (the clustered form; wavefront stacks of one-group layers) */
int vmlmf_stack_dropout_fused(int L, const vmlmf_stack_layer *layers);
/* sizes for the stack: reserve_bytes[l] per layer, one workspace for either direction */
int vmlmf_stack_query(int L, const vmlmf_stack_layer *layers, size_t *reserve_bytes, size_t *workspace_bytes);
/* head (or NULL): a classifier on the TOP layer's final hidden state (Net.lin, vmlmf.py:345,353-355), as in
 * vmlmf_seq_forward_ex / _backward_ex: its logits come out of the forward launch's epilogue, d(hT) = dlogits W enters the
 * backward launch's prologue, dW / db are outputs of the stack's finish launch. */
int vmlmf_stack_forward(int L, const vmlmf_stack_layer *layers, const float *x, const vmlmf_head *head, void *workspace,
                        size_t workspace_bytes, void *stream);
/* dy: gradient of the top layer's y (or NULL); dx: gradient of x (or NULL when not wanted) */
int vmlmf_stack_backward(int L, const vmlmf_stack_layer *layers, const float *x, const float *dy, float *dx,
                         const vmlmf_head *head, void *workspace, size_t workspace_bytes, void *stream);

int vmlmf_seq_backward(const vmlmf_desc *d, const vmlmf_params *p, const float *x, const float *h0,
                       const float *c0, const float *y, const void *reserve, const float *dy,
                       const float *dhT, const float *dcT, float *dx, float *dh0, float *dc0,
                       const vmlmf_grads *g, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Classifier head of the HAR model: logits = h W^T + bias, the nn.Linear(H, 18) applied to the last timestep
 * in Net.forward (V/src/models/vmlmf.py:345,353-355) — SURVEY §8f "next" row.  h: B rows of H floats, row
 * stride ldh (so the last-timestep slice of a (B,T,H) output can be passed without a copy); weight (C,H) and
 * bias (C) in nn.Linear's layout; logits (B,C) dense.  bias may be NULL.
 * Backward: dlogits (B,C) -> dh (B,H dense), dweight (C,H), dbias (C); any of the three may be NULL.  C <= 32.
 * Fixed summation order (no atomics).
 */
int vmlmf_head_forward(int B, int H, int C, const float *h, long long ldh, const float *weight,
                       const float *bias, float *logits, void *stream);
int vmlmf_head_backward(int B, int H, int C, const float *h, long long ldh, const float *weight,
                        const float *dlogits, float *dh, float *dweight, float *dbias, void *stream);

/*
 * Cross-entropy of the classifier logits, mean over the rows whose target != ignore_index: the criterion of
 * the reference's training loop (nn.CrossEntropyLoss on Net's output, V/src/train_test/train.py:58-65) — SURVEY
 * §8f "next" row.  logits (B,C) dense fp32, target (B) int64 in [0, C) or == ignore_index (any other value: the loss
 * becomes NaN, nothing is read out of bounds).  Forward writes the scalar loss, the row
 * log-sum-exps lse (B) and the number of counted rows nvalid (1); backward turns them and the incoming
 * gradient of the loss (device scalar) into dlogits (B,C).  One workgroup in forward: meant for classifier
 * sized problems (the Python wrapper dispatches B*C <= 65536 here and leaves larger ones to the library op).
 * dlogits_unit (B,C), optional: forward also writes the gradient for dloss = 1 there, so a caller that knows its
 * incoming gradient is one (loss.backward() of the training loop) needs no backward launch.
 */
int vmlmf_ce_forward(int B, int C, const float *logits, const int64_t *target, int64_t ignore_index, float *loss,
                     float *lse, float *nvalid, float *dlogits_unit, void *stream);
int vmlmf_ce_backward(int B, int C, const float *logits, const int64_t *target, int64_t ignore_index,
                      const float *lse, const float *nvalid, const float *dloss, float *dlogits, void *stream);

/*
 * Softmax negative log-likelihood of the language-model loop (nll_loss, V/src/train_test/lm_test.py:140-153) —
 * SURVEY §8f rank 3.  scores (R,V) dense fp32 with R = T*B rows, y (R) int64 targets in [0, V) in row order (a target
 * outside that range makes the loss NaN; the reference's indexing raises).
 *   loss = scale * sum_r (logsumexp(scores[r]) - scores[r][y[r]])        (the reference: scale = batch_size / R)
 * Forward writes loss (1), lse (R) and rowloss (R) and reads scores once; backward writes
 * dscores = dloss * scale * (softmax(scores) - onehot(y)) from scores, lse and the device scalar dloss.
 */
int vmlmf_nll_forward(int R, int V, const float *scores, const int64_t *y, float scale, float *loss, float *lse,
                      float *rowloss, void *stream);
int vmlmf_nll_backward(int R, int V, const float *scores, const int64_t *y, float scale, const float *lse,
                       const float *dloss, float *dscores, void *stream);
/* ABI 9, the training form of the same loss: ONE pass over the score matrix.  `scores` (R,V) are the projection's outputs
 * WITHOUT the bias (`bias` (V) or NULL is added here, so the GEMM in front needs no bias epilogue); on return the matrix holds
 * its own gradient for d(loss) = 1, dscores = scale (softmax(scores + bias) - onehot(y)), IN PLACE - the two backward GEMMs of
 * the projection (Linear, V/src/models/vmlmf_lm.py:355-358) consume it where it lies, no second R x V buffer exists - and
 * dbias (V, may be NULL) holds its column sums, the bias gradient.  loss (1), rowloss (R) as above.  scratch:
 * vmlmf_nll_grad_scratch_floats(R, V) floats.  Rows must be 16-byte aligned, V % 4 == 0, V <= 12288 (VMLMF_E_UNSUPPORTED
 * otherwise: use the two calls above).  Fixed summation orders, no atomics. */
size_t vmlmf_nll_grad_scratch_floats(int R, int V);
int vmlmf_nll_forward_grad(int R, int V, float *scores, const float *bias, const int64_t *y, float scale, float *loss,
                           float *rowloss, float *dbias, float *scratch, void *stream);

/*
 * Gradient of the embedding table (Embed, V/src/models/vmlmf_lm.py:46-48: x = w[tokens]; autograd's backward scatter-adds the
 * R = T*B rows of dy (R,H) into a zero (V,H) matrix).  dweight[v] = sum of dy[p] over the positions p with tokens[p] == v, in
 * ascending position order (deterministic, no float atomics); rows no token selects are written as zeros: the call fills all of
 * dweight.  scratch: vmlmf_embed_backward_scratch_bytes(R, V) bytes (one bit per (vocabulary row, position)).  H <= 1024.
 */
size_t vmlmf_embed_backward_scratch_bytes(int R, int V);
/* ---- dropout launches (ABI 11; the scheme: vmlmf_dropout above) ----
 * vmlmf_dropout_fused     1: vmlmf_seq_forward_ex / _backward_ex take extra.drop for this layer (row-block kernels, time-major).
 * vmlmf_dropout_advance   snapshot = state; state.offset += 1 - one tiny launch per training forward, a node of a captured graph
 *                         (every replay draws fresh factors).  state / snapshot: two int64 each, device memory.
 * vmlmf_dropout_apply     y = x * factor over R positions of H columns (x == y allowed): nn.Dropout's forward, and - on the upstream
 *                         gradient with the same (state, site) - its backward.
 * vmlmf_dropout_factors   the factors as a (R, H) tensor, columns mapped as layer `d`'s fused kernels map them (d == NULL: as
 *                         vmlmf_dropout_apply / the embedding entry points do).
 * vmlmf_embed_dropout_forward   out[r] = weight[tokens[r]] * factor (vmlmf_lm.py:434-435 in one pass)
 * vmlmf_embed_dropout_backward  vmlmf_embed_backward on the gradient of that dropped output */
int vmlmf_dropout_fused(const vmlmf_desc *d);
int vmlmf_dropout_advance(int64_t *state, int64_t *snapshot, void *stream);
int vmlmf_dropout_apply(int64_t R, int H, const float *x, float *y, float p, const int64_t *state, int site, void *stream);
int vmlmf_dropout_factors(const vmlmf_desc *d, int64_t R, int H, float p, const int64_t *state, int site, float *factors, void *stream);
int vmlmf_embed_dropout_forward(int R, int H, int V, const int64_t *tokens, const float *weight, float *out, float p,
                                const int64_t *state, int site, void *stream);
int vmlmf_embed_dropout_backward(int R, int H, int V, const int64_t *tokens, const float *dy, float *dweight, void *scratch,
                                 size_t scratch_bytes, float p, const int64_t *state, int site, void *stream);
int vmlmf_embed_backward(int R, int H, int V, const int64_t *tokens, const float *dy, float *dweight, void *scratch,
                         size_t scratch_bytes, void *stream);

/* dst (cols x rows, dense) = src (rows x cols, dense)^T, fp32, out of place.  The LM head's weight gradient dW = dz^T h is fastest as
 * the library GEMM that yields dW^T; this turns it into the (V, H) tensor fc.w.grad is (vmlmf_amd/functional.py: LmHeadLossFn). */
int vmlmf_transpose(int rows, int cols, const float *src, float *dst, void *stream);

/*
 * Optimizer steps of the reference's two training loops, one launch over every parameter tensor.
 * vmlmf_tensor_list carries up to VMLMF_MAX_TENSORS (param, grad, numel, state_offset, step_index) entries; larger
 * models are stepped in several calls.  All tensors fp32, dense.
 *   vmlmf_adam_step      torch.optim.Adam(params, lr) semantics (V/src/train_test/train.py:47,65; betas, eps,
 *                        L2 weight_decay as in torch, no amsgrad).  exp_avg / exp_avg_sq are flat device buffers,
 *                        entry i owns [state_offset[i], state_offset[i] + numel[i]).  `steps` is a device array
 *                        of fp32 step counts (torch counts per parameter); the call increments
 *                        steps[step_index[i]] of every listed tensor first, so it also works inside a hipGraph.
 *   vmlmf_sgd_clip_step  clip_grad_norm_(params, max_norm) followed by param -= lr * grad
 *                        (V/src/train_test/lm_test.py:203-209).  The gradients are scaled in place, the total
 *                        norm before clipping is left in `norm` (device scalar); max_norm <= 0 skips clipping.
 *                        `scratch`: VMLMF_MAX_TENSORS * 64 floats.
 */
#define VMLMF_MAX_TENSORS 48
typedef struct vmlmf_tensor_list {
  void *param[VMLMF_MAX_TENSORS];
  const void *grad[VMLMF_MAX_TENSORS];
  int64_t numel[VMLMF_MAX_TENSORS];
  int64_t state_offset[VMLMF_MAX_TENSORS];
  int32_t step_index[VMLMF_MAX_TENSORS];
  int32_t count;
} vmlmf_tensor_list;
int vmlmf_adam_step(const vmlmf_tensor_list *tensors, float *exp_avg, float *exp_avg_sq, float *steps, float lr,
                    float beta1, float beta2, float eps, float weight_decay, void *stream);
int vmlmf_sgd_clip_step(const vmlmf_tensor_list *tensors, float lr, float max_norm, float *norm, float *scratch,
                        void *stream);
/* ABI 9: non-finite gradients never reach the parameters.  A launch that gave up a bounded wait leaves NaN gradients and the
 * host learns of it one call later (VMLMF_E_PROTOCOL) - inside a replayed hipGraph not at all - so the decision is taken on
 * the device:
 *   vmlmf_adam_step_guarded  `guard`: VMLMF_GUARD_WORDS zero-initialised uint32 device words the caller keeps between steps
 *                            (NULL = vmlmf_adam_step).  When the gradients of the step are not finite the whole step is skipped:
 *                            no step counter ticks, parameters and moments keep their values, guard[VMLMF_GUARD_SKIPPED]
 *                            counts it (guard[VMLMF_GUARD_GO] = 0 for that step).  How "not finite" is found
 *                            (vmlmf_tune("adam_guard", m)): m = 1 (default) - the launch that writes a layer's parameter
 *                            gradients (finish_kernel, inside every backward call of this library) sets a per-device health
 *                            word when one of them is Inf / NaN, and the tick launch of this call reads and clears it: no extra
 *                            launch, no extra pass - covers gradients that came out of this library's backward calls on this
 *                            device since the last guarded step; m = 2 - a gate launch scans every listed gradient (any
 *                            source; one call = one gate: models of more than VMLMF_MAX_TENSORS tensors are gated per call);
 *                            m = 0 - no guard.
 *   vmlmf_sgd_clip_step      skips the step (parameters and gradients untouched) when the total norm is not finite; `norm`
 *                            carries the non-finite value back to the caller. */
#define VMLMF_GUARD_WORDS 72
#define VMLMF_GUARD_GO 64
#define VMLMF_GUARD_SKIPPED 66
int vmlmf_adam_step_guarded(const vmlmf_tensor_list *tensors, float *exp_avg, float *exp_avg_sq, float *steps, float lr,
                            float beta1, float beta2, float eps, float weight_decay, void *guard, void *stream);
/* ABI 10: an optimizer step that takes several calls (more than VMLMF_MAX_TENSORS tensors, several parameter groups) - ONE verdict
 * for all of them.  `flags`: VMLMF_ADAM_FIRST on the first call of the step (it reads the health word and leaves the verdict in the
 * guard block), VMLMF_ADAM_LAST on the last (it clears the health word); the calls of a step share one guard block and one stream.
 * vmlmf_adam_step_guarded is FIRST | LAST: the whole step in one call.  (Before ABI 10 every call read AND cleared the word: the
 * second list of a failed step saw a clean word and applied its NaN gradients.)  With the scanning guard (vmlmf_tune("adam_guard",
 * 2)) every call still gates its own list.  Under the health-word guard a list whose tensors have at most 2^20 elements takes ONE
 * launch (step counters, verdict and update; the last workgroup to finish commits); other lists a tick launch and the update launch. */
#define VMLMF_ADAM_FIRST 1
#define VMLMF_ADAM_LAST 2
int vmlmf_adam_step_ex(const vmlmf_tensor_list *tensors, float *exp_avg, float *exp_avg_sq, float *steps, float lr,
                       float beta1, float beta2, float eps, float weight_decay, void *guard, int flags, void *stream);

/*
 * Data-parallel gradient exchange (no reference line: the reference has no distributed
 * code).  Batch rows are independent through the whole forward and backward, so the ONLY exchange of a training step is the
 * sum over ranks of the parameter gradients; the kernels above already write a layer's gradients into one flat
 * allocation, which is all-reduced IN PLACE on `stream` (RCCL over xGMI; HAR Net: 121 KiB, latency-bound, hence one
 * group call for all buffers of a step and no bucketing).
 *   op: VMLMF_AVG reproduces a mean loss over the global batch (HAR cross-entropy, train.py:63), VMLMF_SUM the LM loss
 *       (mean token NLL x local batch, lm_test.py:147-153).
 *   comm: an RCCL communicator as void*: either one the caller owns (ncclComm_t) or one made here -
 *       rank 0 calls vmlmf_comm_unique_id(), ships the 128 bytes to the other ranks by any means (the Python
 *       binding uses torch.distributed's store), every rank calls vmlmf_comm_init() with its HIP device current.
 * RCCL is bound at run time from the copy the process already holds (PyTorch-ROCm's) or the ROCm install;
 * VMLMF_E_UNSUPPORTED when there is none, VMLMF_E_COMM for RCCL's own errors.
 */
#define VMLMF_SUM 0
#define VMLMF_AVG 1
#define VMLMF_COMM_ID_BYTES 128
int vmlmf_comm_unique_id(void *id128);
int vmlmf_comm_init(void **comm, int world, int rank, const void *id128);
/* ABI 8: the number of ranks RCCL itself reports for the communicator (ncclCommCount): evidence, in a result line, that the
 * exchange really spans N processes */
int vmlmf_comm_count(void *comm, int *ranks);
int vmlmf_comm_destroy(void *comm);
int vmlmf_flat_allreduce(void *buf, size_t n, int op, void *comm, void *stream);
int vmlmf_flat_allreduce_group(int nbuf, void *const *bufs, const size_t *counts, int op, void *comm, void *stream);

/*
 * ABI 13.  One-shot peer-to-peer all-reduce of a SMALL flat fp32 buffer - the HAR network's 121 KiB of gradients between
 * loss.backward() and optimizer.step() (train.py:64-65) - without a collective library: every rank writes
 * its buffer into a slot of every peer's staging area (hipIpc-mapped device memory, one xGMI hop, all peers at once), raises one
 * epoch word per peer, then sums the `world` slots of its own staging area in rank order (the same bits on every rank).  Two
 * launches on the caller's stream (capturable: the epoch lives in device memory), no host work per step.
 *   vmlmf_p2p_create    allocates this rank's staging area for buffers of up to max_floats floats (<= 4 Mi) and returns its IPC
 *                       handle (VMLMF_P2P_HANDLE_BYTES); the caller carries the handles of all ranks to every rank (e.g. an
 *                       all-gather over torch.distributed / MPI; rank order) and calls
 *   vmlmf_p2p_connect   handles = world x VMLMF_P2P_HANDLE_BYTES bytes
 *   vmlmf_p2p_allreduce in place, op VMLMF_SUM / VMLMF_AVG; buf 16-byte aligned, n <= max_floats.  Every rank must call it the same
 *                       number of times.  A peer that never arrives (bounded wait) leaves NaN in buf and the next forward / backward
 *                       entry point returns VMLMF_E_PROTOCOL.
 * At most VMLMF_P2P_MAX_RANKS ranks (one node).  Exercised on hardware only with two processes on ONE device (one GPU per lease):
 * RCCL (vmlmf_flat_allreduce*) stays the default transport of the package.
 */
#define VMLMF_P2P_HANDLE_BYTES 64
#define VMLMF_P2P_MAX_RANKS 8
int vmlmf_p2p_create(void **p2p, int rank, int world, size_t max_floats, unsigned char *handle_out);
int vmlmf_p2p_connect(void *p2p, const unsigned char *handles);
int vmlmf_p2p_allreduce(void *p2p, float *buf, size_t n, int op, void *stream);
int vmlmf_p2p_destroy(void *p2p);

/*
 * Instrumentation for bench.py (roofline leg).  vmlmf_profile_enable(mask): every launch of internal kernel
 * k with bit k set in `mask` is bracketed by a HIP event pair recorded on the SAME stream the kernel is
 * launched on (mask 0 = off, 0xff = all).  vmlmf_profile_read()
 * synchronises the recorded events and returns, per internal kernel, the summed duration in microseconds
 * and the number of launches.  Kernel indices: 0 pack, 1 xproj, 2 rec_fwd, 3 rec_bwd, 4 dqx_dx,
 * 5 wgrad, 6 reduce, 7 finish, 8 head_fwd, 9 head_bwd, 10 ce_fwd, 11 ce_bwd (vmlmf_kernel_name(i) gives the
 * symbol name rocprofv3 reports).
 */
#define VMLMF_NKERNELS 12
int vmlmf_profile_enable(int mask);
int vmlmf_profile_read(float *usec_sum, int32_t *count, int reset);
const char *vmlmf_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* VMLMF_HIP_H */
