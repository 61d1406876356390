// Host side of the C ABI (include/vmlmf_hip.h): descriptor validation, launch geometry, buffer layout,
// and the kernel sequences of one layer's forward / backward.  No torch, no allocation, no sync.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vmlmf_hip.h"
#include "vmlmf_launch.h"

namespace {

thread_local std::string g_err = "";

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// ---- profiling (bench.py): HIP event pairs around every internal launch, on the launch stream ----
constexpr int NKERN = 13;
struct Prof {
  std::mutex mu;
  unsigned mask = 0;  // bit k: bracket kernel k with an event pair
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[NKERN];
} g_prof;

// VMLMF_DEBUG_SYNC=1: synchronise after every internal launch and name it on stderr (finds the kernel behind an
// asynchronous GPU fault; never set in production: it serialises everything and breaks hipGraph capture)
const bool g_debug_sync = getenv("VMLMF_DEBUG_SYNC") != nullptr;
const char* kernel_label(int k);

struct Scope {
  int k;
  hipStream_t s;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  Scope(int which, hipStream_t st) : k(which), s(st) {
    if (g_debug_sync) fprintf(stderr, "[vmlmf] launching %s\n", kernel_label(which));
    if ((g_prof.mask >> which) & 1u) {
      // (profiling instrumentation: an event that could not be made or recorded shows up as a missing / zero sample)
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, s);
    }
  }
  ~Scope() {
    if (g_debug_sync) {
      const hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, "[vmlmf] %s done: %s\n", kernel_label(k), hipGetErrorString(e));
    }
    if (e0 != nullptr) {
      (void)hipEventRecord(e1, s);
      std::lock_guard<std::mutex> lk(g_prof.mu);
      g_prof.ev[k].push_back({e0, e1});
    }
  }
};

long long align64(long long v) { return (v + 63) / 64 * 64; }

// ---- protocol failures inside a launch ----
// The riding weight-gradient workers, the clusters of the row-block kernels and the wavefront hand-overs all wait for other
// workgroups with a bounded number of looks; a wait that gives up leaves NaN in the results (never a plausible wrong number)
// and a code in a status word.  The word lives in host memory mapped into the device (one per device, allocated at the first
// call): the kernel's store costs nothing unless it happens, and the host reads it without a copy or a synchronisation.
// Every forward / backward entry point looks at it first: a failure of an EARLIER launch on the device comes back as
// VMLMF_E_PROTOCOL from the next call (under VMLMF_DEBUG_SYNC from the failing call itself); vmlmf_check_status() after a
// synchronisation tells at once.
constexpr int MAX_DEV = 16;
std::atomic<unsigned*> g_status[MAX_DEV];
std::atomic<bool> g_status_failed[MAX_DEV];   // the allocation itself failed (not: was skipped because of a capture)
std::mutex g_status_mu;
// beside it, in DEVICE memory: the gradient-health word.  finish_kernel sets it when a parameter gradient it writes is not
// finite (the NaN partial products of a launch that gave up a wait); the package's Adam reads it in its tick launch and skips
// that step, then clears it (vmlmf_optim.hip).  Device memory, because the tick launch reads it in every step.
std::atomic<unsigned*> g_health[MAX_DEV];
int g_adam_guard = []() { const char* e = getenv("VMLMF_ADAM_GUARD"); return e ? atoi(e) : 1; }();

// `s`: the stream the caller is about to launch on.  The first call on a device allocates the word; that allocation is not
// capturable, so a first call made while `s` is being captured returns NULL WITHOUT remembering anything (the launch simply
// carries no word; the next call outside a capture allocates it).  Torch captures on a side stream, never the null stream:
// the caller's own stream is what has to be asked.
unsigned* status_word(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  unsigned* w = g_status[dev].load(std::memory_order_acquire);
  if (w != nullptr || g_status_failed[dev].load(std::memory_order_acquire)) return w;
  std::lock_guard<std::mutex> lk(g_status_mu);
  w = g_status[dev].load(std::memory_order_acquire);
  if (w != nullptr || g_status_failed[dev].load(std::memory_order_acquire)) return w;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (cs != hipStreamCaptureStatusNone) return nullptr;          // not now; nothing is latched
  // another thread of the process may be capturing in global mode: the allocation must not invalidate its capture
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  void* p = nullptr;
  const bool ok = hipHostMalloc(&p, 64, hipHostMallocMapped) == hipSuccess && p != nullptr;
  void* hw = nullptr;
  if (ok && hipMalloc(&hw, 256) == hipSuccess && hw != nullptr) {
    if (hipMemset(hw, 0, 256) == hipSuccess) g_health[dev].store((unsigned*)hw, std::memory_order_release);
  } else {
    (void)hipGetLastError();
  }
  if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (ok) {
    memset(p, 0, 64);
    g_status[dev].store((unsigned*)p, std::memory_order_release);
    return (unsigned*)p;
  }
  (void)hipGetLastError();
  g_status_failed[dev].store(true, std::memory_order_release);
  return nullptr;
}
unsigned* health_word(hipStream_t s) {   // allocated together with the status word
  (void)status_word(s);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  return g_health[dev].load(std::memory_order_acquire);
}
// the word if it exists already (host-side readers: never allocates)
unsigned* status_word_if_any() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  return g_status[dev].load(std::memory_order_acquire);
}

// looks a riding weight-gradient worker takes at its rows' progress words before it gives up (vmlmf_tune "test_wride_spin": tests)
constexpr int WRIDE_SPIN_DEFAULT = 1 << 16;
int g_wride_spin = WRIDE_SPIN_DEFAULT;
// set when a worker gave up under the production bound: the workers wait for row workgroups of their own launch, which a GPU
// shared with other processes / launches can keep from getting a CU.  From then on the process takes the
// stand-alone weight-gradient kernel (plan_wride) instead of failing every step; vmlmf_tune("wride", 1) re-arms the riding form.
// (Launches already captured into a hipGraph stay what they are.)
std::atomic<int> g_wride_tripped{0};

const char* status_text(unsigned code) {
  switch (code) {
    case VMLMF_ST_WRIDE: return "a weight-gradient worker riding on the backward launch never saw its rows' progress words (parameter gradients of that call are NaN); the workers wait for workgroups of their own launch and need them resident: when the GPU is shared with other processes or launches that fill its CUs, run with VMLMF_WRIDE=0 (after this report the process does so by itself for eager launches)";
    case VMLMF_ST_CLUSTER: return "a member of a row-block cluster never published its partial (outputs of that call are NaN)";
    case VMLMF_ST_WF_FWD: return "a layer of a wavefront forward launch never received the rows of the layer below (outputs are NaN)";
    case VMLMF_ST_WF_BWD: return "a layer of a wavefront backward launch never received the gradient rows of the layer above (gradients are NaN)";
    case VMLMF_ST_P2P: return "a rank of the peer-to-peer all-reduce never wrote its buffer into this rank's staging area (the reduced buffer is NaN)";
  }
  return "unknown status code";
}

// 0, or VMLMF_E_PROTOCOL with the text of the failure an earlier launch on this device reported (the word is cleared)
int g_tune_generation = 0;                        // bumped by every vmlmf_tune() and by the automatic switch below: kept parameter images / captured graphs of an older one are stale
int take_status() {
  unsigned* w = status_word_if_any();
  if (w == nullptr) return 0;
  const unsigned code = *(volatile unsigned*)w;
  if (code == 0) return 0;
  *(volatile unsigned*)w = 0;
  if (code == VMLMF_ST_WRIDE && g_wride_spin == WRIDE_SPIN_DEFAULT && g_wride_tripped.exchange(1) == 0) ++g_tune_generation;
  return fail(VMLMF_E_PROTOCOL, std::string("an earlier launch on this device gave up a bounded wait: ") + status_text(code));
}
// at the end of an entry point under VMLMF_DEBUG_SYNC: the failure of THIS call
int debug_status(hipStream_t s) {
  if (!g_debug_sync) return 0;
  (void)hipStreamSynchronize(s);
  return take_status();
}

const bool g_xwave = []() {
  const char* e = getenv("VMLMF_XWAVE");
  return e == nullptr || e[0] != '0';
}();

// wgrad chunking (A/B measurements): VMLMF_WCHUNKS = target number of row chunks, VMLMF_WMIN = fewest rows per chunk
// (non-numeric or non-positive values fall back to the defaults: a zero here would divide by zero in make_geo)
int env_pos(const char* name, int dflt) {
  const char* e = getenv(name);
  if (e == nullptr) return dflt;
  const int v = atoi(e);
  return v >= 1 ? v : dflt;
}
int env_int(const char* name, int dflt) {   // any integer (switches with a -1 / 0 / 1 meaning)
  const char* e = getenv(name);
  return e == nullptr ? dflt : atoi(e);
}
const int g_wchunks = env_pos("VMLMF_WCHUNKS", 64);
const int g_rc = env_pos("VMLMF_RC", 0);   // dqx_dx rows per workgroup (A/B); 0 = derived from the row count
const int g_wmin = env_pos("VMLMF_WMIN", 64);   // config C (3072 rows): 0.2546 ms at 32 or 48, 0.2428 at 64, 0.243 at 96
// weight-gradient workers riding on rec_bwd_kernel's launch (vmlmf_atb.inc): VMLMF_WRIDE=0 off; VMLMF_WRIDE_K = workers per
// task, VMLMF_WRIDE_MAXB = largest batch that rides (beyond it the rows fill the chip and the workers only compete with them)
const bool g_wride = []() {
  const char* e = getenv("VMLMF_WRIDE");
  return e == nullptr || e[0] != '0';
}();
const int g_wride_k = env_pos("VMLMF_WRIDE_K", 32);
const int g_wride_maxb = env_pos("VMLMF_WRIDE_MAXB", 64);   // round 3, rec3_bwd_kernel rows (61 us alone): ride on / off at H = 180, T = 128: B 32 0.158 / 0.165 ms, 64 0.160 / 0.173, 72 0.185 / 0.178, 80 0.187 / 0.177, 96 0.192 / 0.181 (the faster rows outrun the workers once fewer than ~180 CUs are left for them; round 2, slower rows: rode up to 96)
const int g_wride_lag = env_pos("VMLMF_WRIDE_LAG", 3);
const int g_wride_rc = env_pos("VMLMF_WRIDE_RC", 32);
// looks a riding worker takes at the progress words before it gives up; vmlmf_tune("test_wride_spin", n) shortens it so that
// tests can provoke the failure path (NaN gradients + VMLMF_E_PROTOCOL) on purpose
int g_cus[MAX_DEV] = {0};   // compute units of the device (hipDeviceProp_t::multiProcessorCount), looked up once
int device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
  if (g_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_cus[dev] = n;
  }
  return g_cus[dev];
}

// Row-block MFMA kernels (vmlmf_rb.hip): -1 = automatic (large batches, and layers beyond the register-resident VALU kernels),
// 0 = never, 1 = wherever an instantiation exists.  VMLMF_RB in the environment, or vmlmf_tune("rb", v) at run time.
int g_rb_mode = []() { const char* e = getenv("VMLMF_RB"); return e ? atoi(e) : -1; }();
// automatic mode: batch rows from which the row-block kernels take over on layers the VALU kernels cover as well.  0 = never:
// measured on MI355X the one-row-per-CU kernels win at every batch size up to 2048 - sixteen rows' tape
// traffic through ONE CU's memory pipe (~25 GB/s) costs more than the MFMAs save
int g_rb_minB = env_pos("VMLMF_RB_MINB", 0);
int g_rb_S = env_pos("VMLMF_RB_S", 0);            // cluster size for large layers (0 = the smallest that has an instantiation)
int g_rb_rows = env_pos("VMLMF_RB_ROWS", 0);      // live batch rows per workgroup: 16 / 8 / 4 (0 = automatic)
// third form of the recurrent kernels (vmlmf_rec3.inc) where it covers the layer: VMLMF_REC3=0 / vmlmf_tune("rec3", 0) keeps
// rec_fwd_kernel / rec_bwd_kernel (A/B runs); bit 1 = forward always, bit 2 = backward, bit 4 = forward when there are more
// batch rows than CUs (rec3_fwd_kernel needs ~170 VGPRs, two workgroups share a CU; rec_fwd_kernel's x-projection wave needs
// 256, so its workgroups run in rounds: measured B = 512 138 -> 104 us, 0.402 -> 0.370 ms per step; up to B = 256 the two tie)
int g_rec3 = []() { const char* e = getenv("VMLMF_REC3"); return e ? atoi(e) : 6; }();
// fourth form of the backward (vmlmf_rec4.inc: weight gradients formed inside the rows' workgroups, no dpre tape, no
// weight-gradient launch): VMLMF_INROW / vmlmf_tune("inrow", v): 0 = never, 1 = wherever it covers the layer, -1 = automatic
// (batches beyond the riding workers' range)
int g_inrow = []() { const char* e = getenv("VMLMF_INROW"); return e ? atoi(e) : -1; }();
// wgrad_ring_kernel for the weight gradients of large layers: -1 automatic (the step-wise / clustered layers), 0 never, 1 wherever
// it applies (VMLMF_WRING / vmlmf_tune("wring", v))
int g_wring = []() { const char* e = getenv("VMLMF_WRING"); return e ? atoi(e) : -1; }();

// ---- geometry ----
// force_W: at least this many waves of hidden units per group (a stack whose layers differ in hidden_size runs every layer on the
// widest one's thread-slot geometry: the surplus slots are padding, as the slots behind a hidden size that is no multiple of 64 are)
int make_geo(const vmlmf_desc* d, VGeo* out, RbGeo* rbout = nullptr, int force_W = 0) {
  if (d == nullptr) return fail(VMLMF_E_BADARG, "null descriptor");
  VGeo g;
  memset(&g, 0, sizeof(g));
  g.variant = d->variant;
  g.B = d->B;
  g.T = d->T;
  g.I = d->I;
  g.H = d->H;
  g.rw = d->w_rank;
  if (d->dtype != VMLMF_DT_F32 && d->dtype != VMLMF_DT_BF16) return fail(VMLMF_E_BADARG, "dtype must be VMLMF_DT_F32 or VMLMF_DT_BF16");
  g.bf = d->dtype == VMLMF_DT_BF16 ? 1 : 0;
  if (g.variant < 1 || g.variant > 6) return fail(VMLMF_E_BADARG, "variant must be 1..6");
  if (g.B < 1 || g.T < 1 || g.I < 1 || g.H < 1 || g.rw < 1)
    return fail(VMLMF_E_BADARG, "B, T, I, H, w_rank must be positive");
  const bool grouped = g.variant == VMLMF_V2_GROUP_CELL || g.variant == VMLMF_V4_LM_GROUP ||
                       g.variant == VMLMF_V6_GROUP_NOVM;
  const bool lm = g.variant == VMLMF_V3_LM || g.variant == VMLMF_V4_LM_GROUP;
  g.novm = (g.variant == VMLMF_V5_LMF_CELL || g.variant == VMLMF_V6_GROUP_NOVM) ? 1 : 0;
  g.pergate = g.variant == VMLMF_V5_LMF_CELL ? 1 : 0;
  g.xperm = g.variant == VMLMF_V6_GROUP_NOVM ? 1 : 0;
  g.G = grouped ? d->g : 1;
  if (grouped && g.G < 1) return fail(VMLMF_E_BADARG, "g must be positive");
  if (g.G > VMLMF_MAX_G)
    return fail(VMLMF_E_UNSUPPORTED, "g > 2 is not covered by the HIP kernels (the reference never builds it)");
  if (g.H % g.G != 0) return fail(VMLMF_E_SHAPE, "hidden_size must be divisible by g (vmlmf_group.py:73)");
  // the reference fails on these shapes too (vmlmf.py:94 / vmlmf_lm.py:243)
  if (!lm && !g.novm && g.I > g.H)
    return fail(VMLMF_E_SHAPE, "input_size > hidden_size: the reference cell raises (vmlmf.py:94,103)");

  if (lm && g.I != g.H) return fail(VMLMF_E_SHAPE, "LM layers need input_size == hidden_size (vmlmf_lm.py:243)");
  g.ru0 = d->u_ranks[0];
  g.ru1 = g.G == 2 ? d->u_ranks[1] : 0;
  if (g.ru0 < 1 || (g.G == 2 && g.ru1 < 1)) return fail(VMLMF_E_BADARG, "u_ranks must be positive");
  g.Hg = g.H / g.G;
  g.W = (g.Hg + 63) / 64;
  if (force_W > g.W) g.W = force_W;
  g.NW = g.G * g.W;
  g.NT = g.NW * 64;
  g.off1 = vg_pad8(g.ru0);
  g.KH = g.off1 + (g.G == 2 ? vg_pad8(g.ru1) : 0);
  g.KX = vg_pad8(g.rw);
  g.NP = (g.KH + 15) / 16;
  g.KQ = g.NP * 16;
  g.NPX = (g.KX + 15) / 16;
  g.KQX = g.NPX * 16;
  g.flat = g.variant == VMLMF_V4_LM_GROUP ? 1 : 0;
  g.hperm = (g.variant == VMLMF_V2_GROUP_CELL || g.variant == VMLMF_V6_GROUP_NOVM) ? 1 : 0;
  g.time_major = d->time_major ? 1 : 0;
  g.training = d->training ? 1 : 0;
  if (g.time_major) {
    g.sxT = (long long)g.B * g.I;
    g.sxB = g.I;
    g.syT = (long long)g.B * g.H;
    g.syB = g.H;
  } else {
    g.sxT = g.I;
    g.sxB = (long long)g.T * g.I;
    g.syT = g.H;
    g.syB = (long long)g.T * g.H;
  }
  if (g.KX > 32) return fail(VMLMF_E_UNSUPPORTED, "padded w_rank > 32 is not covered by the HIP kernels");
  if (g.G * g.KH > 128) return fail(VMLMF_E_UNSUPPORTED, "padded hidden rank (summed over groups) > 128 is not covered");
  // register-resident persistent kernels need <= 32 ranks per unit and <= 512 thread slots; larger layers
  // (e.g. H = 650, ranks [32,32]) run the step-wise path of vmlmf_generic.hip
  // (the cells without vm accept input_size > hidden_size; the persistent kernels keep x-side quantities in the slots of the
  // first I units, so such a layer takes the step-wise path, whose x side is indexed by input)
  g.generic = (g.KH > 32 || g.NT > 512 || g.I > g.H) ? 1 : 0;
  // One batch row per workgroup, whatever the batch: with more rows than CUs the workgroups queue up, which
  // measured at least as fast as two rows per workgroup at every size (H = 180, T = 128: B = 512 0.49 vs 0.55 ms,
  // 768 0.71 vs 0.78, 1024 0.98 vs 0.96, 2048 1.84 vs 2.00); the kernels keep their R template parameter.
  g.R = 1;
  g.nwg = (g.B + g.R - 1) / g.R;
  g.Bp = g.nwg * g.R;
  if ((long long)g.T * g.Bp * g.NT * 4 >= (1LL << 31) || (long long)g.T * g.B * g.H >= (1LL << 31))
    return fail(VMLMF_E_UNSUPPORTED, "T*B*H too large for the 32-bit element offsets of the kernels");
  // backward, parallel part: dqx_dx works on RC rows per workgroup (groups of 8 per barrier); wgrad keeps
  // register accumulators over RC2 rows per chunk, 5 parts per chunk
  const int TB = g.T * g.B;
  // (measured at the headline shape: ~1024 dqx_dx workgroups 12.9 us, 512 14.0, 256 16.7; 64 wgrad chunks:
  // 128 chunks gain 2 us there and lose them again in reduce_cg_kernel, 32 chunks cost 10 us)
  int rc = ((TB + 1023) / 1024 + 7) / 8 * 8;
  if (rc < 8) rc = 8;
  if (g_rc > 0) rc = g_rc;
  g.RC = rc;
  g.nblk = (TB + rc - 1) / rc;
  int rc2 = (TB + g_wchunks - 1) / g_wchunks;
  if (rc2 < g_wmin) rc2 = g_wmin;
  g.RC2 = rc2;
  g.nchunk = (TB + rc2 - 1) / rc2;
  g.foldx = (!g.generic && g.I <= g.KX) ? 1 : 0;
  g.NA = 5 * g.KX + 5 * g.KH + 12;
  // row-block MFMA recurrence?
  {
    RbGeo q;
    memset(&q, 0, sizeof(q));
    g.rb = 0;
    if (g.bf) {   // the bf16 variant IS the row-block family
      if (!g.generic && g.I <= g.H && rb_geometry(g, 1, &q, g_rb_rows)) g.rb = 1;
      else return fail(VMLMF_E_UNSUPPORTED, "dtype bf16: implemented by the row-block MFMA kernels for one-group layers (V1, V3, V5) "
                                            "with padded rank <= 32 and <= 512 thread slots");
    } else if (g_rb_mode != 0 && g.I <= g.H) {
      if (g.generic) {          // factors beyond one CU's registers: a cluster of S workgroups per 16-row block
        // measured at H = 650, B = 256 (members of a cluster on one XCD): group layer (ranks 32+32) 2.98 / 2.21 / 2.09 ms with
        // clusters of 4 / 8 / 16, plain layer (rank 32) 1.84 / 1.64 / 1.60 ms: the largest cluster first
        const int cand[] = {g_rb_S, 16, 8, 4, 2};
        for (int i = (g_rb_S > 0 ? 0 : 1); i < 5 && g.rb == 0; ++i)
          if (cand[i] > 1 && rb_geometry(g, cand[i], &q, g_rb_rows)) g.rb = cand[i];
      } else if ((g_rb_mode == 1 || (g_rb_minB > 0 && g.B >= g_rb_minB)) && rb_geometry(g, 1, &q, g_rb_rows)) {
        g.rb = 1;
      }
    }
    if (g.rb != 0) g.foldx = 0;   // the x-fold belongs to the x-projection wave's layers; here qx always exists
    if (rbout != nullptr) *rbout = q;
  }
  {
    const long long GK = (long long)g.G * g.KH;
    const long long nb1 = (vg_nb1(g) + 31) / 32 * 32, nb2 = (GK + 31) / 32 * 32, nb3 = (g.KX + 31) / 32 * 32;
    g.PCH = (long long)g.NT * 4 * nb1 + (long long)((g.H + 31) / 32) * 32 * nb2 +
            (long long)((g.I + 31) / 32) * 32 * nb3 + 3LL * g.NT * 4;
  }
  *out = g;
  return 0;
}

// ---- buffer layouts (float offsets) ----
struct Layout {
  // reserve (training) : PACK | qx | gates | cs | Qs
  long long r_pack, r_qx, r_gates, r_cs, r_Qs, r_prog, r_total;
  // forward workspace  : PACK (inference only) | gx
  long long f_pack, f_gx, f_qx, f_trash, f_Qtmp, f_P, f_ccar, f_zeros, f_part, f_xq, f_flag, f_total;
  // backward workspace : dpre | dQs | wpart | cgrad
  long long b_dpre, b_dQs, b_dqx, b_wpart, b_cgrad, b_trash, b_dHrec, b_ehterm, b_dcar, b_part, b_xq, b_flag, b_headdh, b_dux, b_total;
};

Layout make_layout(const VGeo& g, const VPack& P, const RbGeo& q) {
  Layout L;
  const long long TB = (long long)g.T * g.B;
  const long long TS = (long long)g.T * g.Bp * g.NT;   // (t, padded row, thread slot)
  long long o = 0;
  L.r_pack = o, o += align64(P.total);
  // (16 spare rows behind qx, Q, dQ, dqx of the layers wgrad_ring_kernel may take: its last stage fetches whole 16-row pieces)
  const long long TBp = TB + (g.NT >= 256 ? 16 : 0);
  L.r_qx = o, o += align64(TBp * g.KX);
  L.r_gates = o, o += align64(TS * 4);
  L.r_cs = o, o += align64(TS + (long long)g.Bp * g.NT);   // slice 0 = c0, slice t+1 = c_t
  L.r_Qs = o, o += align64(TBp * g.G * g.KH);
  L.r_prog = o, o += align64((long long)g.B * WR_PROG_STRIDE);   // progress words of the backward rows (zero between launches: cleared by the forward kernel)
  L.r_total = o;
  o = 0;
  L.f_pack = o, o += align64(P.total);
  L.f_gx = o, o += align64(TS * 4);
  L.f_qx = o, o += align64(g.generic ? TB * g.KX : 0);   // qx of an inference call on a large layer (the MFMA x-side expansion reads it)
  L.f_trash = o, o += 64;
  {
    const long long gen = (g.generic && !g.rb) ? 1 : 0, BN = (long long)g.B * g.NT;
    L.f_Qtmp = o, o += align64(gen * g.B * g.G * g.KH);
    L.f_P = o, o += align64(gen * BN * 4);
    L.f_ccar = o, o += align64(gen * BN);
    L.f_zeros = o, o += align64(gen * (long long)g.B * g.H);
    L.f_part = o, o += align64(gen * (long long)VG_GEMM_SPLIT * ((g.B + 63) / 64 * 64) * ((g.G * g.KH + 63) / 64 * 64));
  }
  L.f_xq = o, o += align64(g.rb ? q.xq_floats : 0);
  L.f_flag = o, o += align64(g.rb ? q.flag_words : 0);
  L.f_total = o;
  o = 0;
  L.b_dpre = o, o += align64(TS * 4);
  L.b_dQs = o, o += align64(TBp * g.G * g.KH);
  L.b_dqx = o, o += align64(TBp * g.KX);
  // (one partial block per chunk of rows, or - backward with the weight gradients formed in the rows' workgroups - per workgroup)
  {
    long long blocks = g.nchunk;
    if (!g.generic && !g.rb && rec4_bwd_supported(g) && g.nwg > blocks) blocks = g.nwg;
    L.b_wpart = o, o += align64(blocks * g.PCH);
  }
  L.b_cgrad = o, o += align64((long long)g.NA * g.NT + (g.I > g.H ? (long long)g.I * g.KX : 0));   // + dU_x by input when I > H
  L.b_trash = o, o += 64;
  {
    const long long gen = (g.generic && !g.rb) ? 1 : 0, BN = (long long)g.B * g.NT;
    L.b_dHrec = o, o += align64(gen * (long long)g.B * g.H);
    L.b_ehterm = o, o += align64(gen * BN);
    L.b_dcar = o, o += align64(gen * BN);
    // (the dqx product of large layers keeps its split-K scratch under the row-block kernels too)
    L.b_part = o, o += align64((g.generic ? 1 : 0) * (long long)VG_GEMM_SPLIT * ((g.B + 63) / 64 * 64) * ((g.G * g.KH + 63) / 64 * 64));
  }
  L.b_xq = o, o += align64(g.rb ? q.xq_floats : 0);
  L.b_flag = o, o += align64(g.rb ? q.flag_words : 0);
  L.b_headdh = o, o += align64((g.rb || g.generic) ? (long long)g.B * g.H : 0);   // d(hT) of a classifier on the row-block / step-wise families
  // the riding workers' shares of d(u_x) (finish2_kernel): [worker index][task][16 x 16]
  L.b_dux = o, o += align64((!g.generic && !g.rb && finish2_ok(g)) ? (long long)g.nchunk * (g.NT / 8) * 256 : 0);
  L.b_total = o;
  return L;
}

RefP to_refp(const vmlmf_params* p) {
  RefP r;
  r.dia_x = p->dia_x, r.dia_h = p->dia_h, r.u_x = p->u_x, r.v_x = p->v_x, r.b_x = p->b_x, r.b_h = p->b_h;
  r.u_h0 = p->u_h[0], r.u_h1 = p->u_h[1], r.v_h0 = p->v_h[0], r.v_h1 = p->v_h[1];
  for (int k = 0; k < 4; ++k) r.wg[k] = p->w_gate[k], r.ug[k] = p->u_gate[k], r.bg[k] = p->b_gate[k];
  return r;
}

// vmlmf_params and vmlmf_grads have the same members; one check serves both
template <class P>
int check_pointers(const VGeo& g, const P* p, const char* what) {
  if (p == nullptr) return fail(VMLMF_E_BADARG, std::string("null ") + what);
  bool ok = p->u_x && p->u_h[0];
  if (g.pergate) {
    for (int k = 0; k < 4; ++k) ok = ok && p->w_gate[k] && p->u_gate[k] && p->b_gate[k];
  } else {
    ok = ok && p->v_x && p->b_x && p->b_h && p->v_h[0];
    if (!g.novm) ok = ok && p->dia_x && p->dia_h;
  }
  if (!ok) return fail(VMLMF_E_BADARG, std::string("null pointer in ") + what);
  if (g.G == 2 && (!p->u_h[1] || !p->v_h[1]))
    return fail(VMLMF_E_BADARG, std::string(what) + ": group variant needs u_h[1], v_h[1]");
  return 0;
}

int check_params(const VGeo& g, const vmlmf_params* p) { return check_pointers(g, p, "params"); }

// Direct mode (vmlmf_direct.inc): the recurrent kernels of a layer build their register images from the reference layouts in their
// own prologues and pack_kernel leaves the call.  VMLMF_DIRECT=0 / vmlmf_tune("direct", 0): always pack (A/B runs).  The forward and
// the backward of a call pair must reach the same verdict: it depends on the descriptor, the parameter addresses and the kernel
// selection only.
int g_direct = []() { const char* e = getenv("VMLMF_DIRECT"); return e ? atoi(e) : 1; }();
// finish2_kernel behind a backward with riding workers (one launch instead of reduce_cg_kernel + finish_kernel): VMLMF_FINISH2=0 /
// vmlmf_tune("finish2", 0) keeps the two launches (A/B runs)
int g_finish2 = []() { const char* e = getenv("VMLMF_FINISH2"); return e ? atoi(e) : 1; }();
bool uses_rec3_fwd(const VGeo& g) {
  return g_xwave && vg_xwave_ok(g) && ((g_rec3 & 1) || ((g_rec3 & 4) && g.nwg > device_cus())) && rec3_fwd_supported(g);
}
bool direct_ok(const VGeo& g, const vmlmf_params* p) {
  if (g_direct == 0 || !(g.variant == VMLMF_V1_CELL || g.variant == VMLMF_V3_LM)) return false;
  if (g.generic || g.rb || g.bf || g.G != 1 || g.R != 1 || !g.foldx || !(g_xwave && vg_xwave_ok(g)) || uses_rec3_fwd(g)) return false;
  if (!(g.KH == 8 || g.KH == 16) || g.ru0 != g.KH || !(g.KX == 8 || g.KX == 16) || g.rw != g.KX) return false;
  // a training call's backward must be one of the kernels that can do the same (rec3_bwd_kernel / rec4_bwd_kernel)
  if (g.training && !((g_rec3 & 2) && rec3_bwd_supported(g))) return false;   // (its backward: rec3_bwd_kernel / rec4_bwd_kernel)
  const uintptr_t al = (uintptr_t)p->v_h[0] | (uintptr_t)p->u_h[0] | (uintptr_t)p->v_x | (uintptr_t)p->u_x;
  return (al & 15u) == 0;   // rows are read as 16-byte loads
}

int hip_fail(int rc, const char* what) {
  if (rc == 0) return 0;
  if (rc == -3) return fail(VMLMF_E_UNSUPPORTED, std::string(what) + ": no kernel instantiation for this geometry");
  return fail(rc, std::string(what) + ": " + hipGetErrorString((hipError_t)rc));
}

// the batched half of a layer's backward: every weight gradient (MFMA products over all rows), their fixed-order sum, and
// the reference-layout gradients
static WghArgs wgrad_args(const Layout& L, const float* x, const float* y, const float* h0, const float* rs, float* ws) {
  WghArgs wh;
  wh.dpre = ws + L.b_dpre, wh.x = x, wh.y = y, wh.h0 = h0, wh.qx = rs + L.r_qx, wh.dqx = ws + L.b_dqx;
  wh.Qs = rs + L.r_Qs, wh.dQs = ws + L.b_dQs, wh.wpart = ws + L.b_wpart;
  return wh;
}

// Do the weight-gradient products ride on the recurrent backward launch?  Layers of the persistent VALU kernels whose x-side
// gradient folds into the dpre product (no dqx operand, which only exists after that launch), with few enough batch rows that
// most of the chip is idle during the recurrence.  Fills w (K = 0: no).
static void plan_wride(const VGeo& g, const Layout& L, const float* x, const float* y, const float* h0, const float* rs, float* ws,
                       WRide* w, hipStream_t s, const float* vx = nullptr) {
  memset(w, 0, sizeof(*w));
  const int n1 = (vg_nb1(g) + 31) / 32, n2 = (g.G * g.KH + 31) / 32;
  // (g.flat: a V4 layer small enough for the x-fold - hidden_size <= 16 - takes the stand-alone weight-gradient kernel: the riding
  //  instantiations of the flat layout left the library in round 5 as unreachable, and such a layer's backward was refused since -
  //  found by tools/fuzz_parity.py in round 6)
  if (!g_wride || g_wride_tripped.load() != 0 || g.rb || g.generic || g.bf || !g.foldx || g.flat || g.R != 1 || g.NT > 256 || g.B > g_wride_maxb || n1 > 2 || n2 > 2) return;
  const WghArgs wh = wgrad_args(L, x, y, h0, rs, ws);
  w->a.dpre = wh.dpre, w->a.x = wh.x, w->a.y = wh.y, w->a.h0 = wh.h0, w->a.qx = wh.qx, w->a.dqx = wh.dqx, w->a.Qs = wh.Qs;
  w->a.dQs = wh.dQs, w->a.P = wh.wpart;
  w->prog = reinterpret_cast<unsigned*>(const_cast<float*>(rs + L.r_prog));
  // rows per chunk: a part of a step's batch rows when they divide evenly (one batch of loads per chunk: the last chunk's
  // latency is the tail of the launch), else whole steps of at least 64 rows
  if (g.B % g_wride_rc == 0 && g_wride_rc % 2 == 0) w->S = g_wride_rc;
  else w->S = g.B * (g.B >= 64 ? 1 : (64 + g.B - 1) / g.B);
  const int nck = (g.T * g.B + w->S - 1) / w->S;
  int K = g_wride_k < g.nchunk ? g_wride_k : g.nchunk;         // partial blocks: the workspace holds nchunk of them
  K = K < nck ? K : nck;
  w->tasks = g.NT / 8 + (g.H + 31) / 32;
  const int wpw = (g.NT + 128) / 64;
  w->ntg = (w->tasks + wpw - 1) / wpw;
  // every workgroup of the launch has a CU of its own (the launch asks for more than half a CU's LDS): rows + workers must
  // fit the chip at once, or the workers behind the last CU would only start when the others have finished
  // (the CU count of THIS device, less a margin of eight for whatever else is resident: on a partitioned or masked device a
  // fixed 248 would queue workers behind the rows, and a queued worker can only give up)
  const int room = (device_cus() - 8 - g.nwg) / w->ntg;
  K = K < room ? K : room;
  if (K < 4) return;
  w->K = K;
  w->lag = g_wride_lag < 8 ? g_wride_lag : 8;
  w->spin = (unsigned)g_wride_spin;
  w->status = status_word(s);
  // one launch behind this one finishes every gradient (finish2_kernel) where it covers the layer: the workers then contract their
  // x-fold tiles with v_x themselves
  if (g_finish2 != 0 && finish2_ok(g) && vx != nullptr) w->dux = ws + L.b_dux, w->vx = vx;
}

static int backward_tail(const VGeo& g, const Layout& L, const vmlmf_params* p, const vmlmf_grads* gr, const float* x, const float* y,
                         const float* h0, const float* rs, float* ws, const HeadBwd& hb, hipStream_t s, const WRide* ride = nullptr,
                         const int inrow_blocks = 0) {
  int rc;
  const WghArgs wh = wgrad_args(L, x, y, h0, rs, ws);
  const bool rode = ride != nullptr && ride->K > 0;
  int ring_nc[3] = {0, 0, 0};
  if (!rode && inrow_blocks == 0) {
    Scope sc(5, s);
    // large layers: operands through an LDS ring, long chunks (vmlmf_wgrad_ring.hip); -1: where it was measured faster
    const bool ring = g_wring != 0 && wgrad_ring_ok(g) && (g_wring > 0 || (g.generic && (long long)g.T * g.B >= 1024));
    int rr = ring ? launch_wgrad_ring(g, wh, device_cus(), ring_nc, s) : -3;
    if (rr == -3) {   // not taken, or no instantiation / no LDS for it on this device: the stand-alone products, one chunking for all
      ring_nc[0] = ring_nc[1] = ring_nc[2] = 0;
      rr = launch_wgrad_h(g, wh, s);
    }
    if ((rc = hip_fail(rr, "wgrad")) != 0) return rc;
  }
  RefG og;
  og.dia_x = gr->dia_x, og.dia_h = gr->dia_h, og.u_x = gr->u_x, og.v_x = gr->v_x, og.b_x = gr->b_x;
  og.b_h = gr->b_h, og.u_h0 = gr->u_h[0], og.u_h1 = gr->u_h[1], og.v_h0 = gr->v_h[0], og.v_h1 = gr->v_h[1];
  for (int k = 0; k < 4; ++k) og.wg[k] = gr->w_gate[k], og.ug[k] = gr->u_gate[k], og.bg[k] = gr->b_gate[k];
  if (rode && ride->dux != nullptr) {   // the riding workers left their d(u_x) shares: ONE launch sums the K blocks and finishes
    Scope sc(12, s);
    return hip_fail(launch_finish2(g, to_refp(p), ws + L.b_wpart, ride->dux, ride->K, og, hb, ride->prog, s, health_word(s)), "finish2");
  }
  {
    Scope sc(6, s);
    VGeo gr_ = g;
    if (rode) gr_.nchunk = ride->K;   // one partial block per worker index; the progress words go back to zero here
    if (inrow_blocks > 0) gr_.nchunk = inrow_blocks;   // one partial block per workgroup of rec4_bwd_kernel
    if ((rc = hip_fail(launch_reduce(gr_, ws + L.b_wpart, ws + L.b_cgrad, rode ? ride->prog : nullptr, s,
                                     ReduceCounts{{ring_nc[0], ring_nc[1], ring_nc[2]}}), "reduce")) != 0) return rc;
  }
  {
    Scope sc(7, s);
    if ((rc = hip_fail(launch_finish(g, to_refp(p), ws + L.b_cgrad, og, hb, s, health_word(s)), "finish")) != 0) return rc;
  }
  return 0;
}

}  // namespace

unsigned* vmlmf_health_word_if_any() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  return g_health[dev].load(std::memory_order_acquire);
}
int vmlmf_adam_guard_mode() { return g_adam_guard; }

// error text for the other translation units of the C ABI (vmlmf_comm.cpp)
int vmlmf_set_error(int code, const std::string& msg) { return fail(code, msg); }
unsigned* vmlmf_status_word(void* stream) { return status_word((hipStream_t)stream); }   // (vmlmf_p2p.hip)

extern "C" {

int vmlmf_abi_version(void) { return VMLMF_ABI_VERSION; }

const char* vmlmf_build_info(void) { return "vmlmf_hip gfx950 fp32 persistent-rnn (register-resident U/V, DPP rank reduce)"; }

const char* vmlmf_last_error(void) { return g_err.c_str(); }

int vmlmf_query(const vmlmf_desc* d, vmlmf_sizes* out) {
  if (out == nullptr) return fail(VMLMF_E_BADARG, "null sizes");
  VGeo g;
  RbGeo q;
  const int rc = make_geo(d, &g, &q);
  if (rc != 0) return rc;
  const VPack P = vg_pack_layout(g, q.total);
  const Layout L = make_layout(g, P, q);
  const long long ws = L.f_total > L.b_total ? L.f_total : L.b_total;
  out->workspace_bytes = (size_t)ws * sizeof(float);
  out->reserve_bytes = (size_t)L.r_total * sizeof(float);
  out->rows_per_wg = g.rb ? q.rbl : g.R;
  out->threads_per_wg = g.rb ? 256 : g.NT;
  out->workgroups = g.rb ? q.nrb * q.S : g.nwg;
  out->kx = g.KX;
  out->kh = g.KH;
  return 0;
}

// ---- parameter images kept by the caller (vmlmf_pack_params / *_packed) ----
// An image is only valid for the geometry and the kernel selection (vmlmf_tune generation) it was packed for, and it lives in
// device memory, which the forward / backward calls never read back.  So the signature is kept ON THE HOST: vmlmf_pack_params
// records (device address -> signature) in a small registry, and a *_packed call whose image is unknown, or was packed for
// another geometry or under another kernel selection, returns VMLMF_E_BADARG instead of running kernels on a foreign layout
// (verdict r3: the header that used to sit in front of the image was written and never checked).  The registry holds the last
// PK_REG images per process; an image that fell out has to be packed again.  The 256 bytes in front of the image stay
// reserved (alignment of the images behind them).
constexpr int PK_HDR = 64;   // floats
struct PackSig {
  long long v[10];
  bool operator==(const PackSig& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
};
static PackSig pack_signature(const VGeo& g, const VPack& P, const RbGeo& q) {
  PackSig s;
  const long long v[10] = {g.variant, P.total, g.rb, g.generic * 2 + g.bf, g.NT, (long long)g.KH * 1000 + g.KX, q.total,
                           (long long)g.I * 100000 + g.H, (long long)g.G * 100000 + g.ru0 * 100 + g.ru1, g_tune_generation};
  memcpy(s.v, v, sizeof(v));
  return s;
}
constexpr int PK_REG = 256;
struct PackReg {
  std::mutex mu;
  const void* ptr[PK_REG] = {nullptr};
  PackSig sig[PK_REG];
  int next = 0;
  void put(const void* p, const PackSig& s) {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < PK_REG; ++i)
      if (ptr[i] == p) { sig[i] = s; return; }
    ptr[next] = p, sig[next] = s;
    next = (next + 1) % PK_REG;
  }
  // 0 = matches, 1 = unknown image, 2 = packed for something else
  int check(const void* p, const PackSig& s) {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < PK_REG; ++i)
      if (ptr[i] == p) return sig[i] == s ? 0 : 2;
    return 1;
  }
} g_packreg;
static int check_packed(const void* packed, const VGeo& g, const VPack& P, const RbGeo& q) {
  switch (g_packreg.check(packed, pack_signature(g, P, q))) {
    case 0: return 0;
    case 1: return fail(VMLMF_E_BADARG, "packed: not an image vmlmf_pack_params made at this address (or it left the registry of the last 256 images: pack it again)");
  }
  return fail(VMLMF_E_BADARG, "packed: the image was packed for another descriptor or under another kernel selection (vmlmf_tune): pack it again");
}

int vmlmf_pack_bytes(const vmlmf_desc* d, size_t* bytes) {
  if (bytes == nullptr) return fail(VMLMF_E_BADARG, "null size");
  VGeo g;
  RbGeo q;
  const int rc = make_geo(d, &g, &q);
  if (rc != 0) return rc;
  if (g.generic) return fail(VMLMF_E_UNSUPPORTED, "kept parameter images: not for the step-wise / clustered layers (their image carries per-call state)");
  const VPack P = vg_pack_layout(g, q.total);
  *bytes = sizeof(float) * (size_t)(PK_HDR + P.total);
  return 0;
}

int vmlmf_pack_params(const vmlmf_desc* d, const vmlmf_params* p, void* packed, void* stream) {
  VGeo g;
  RbGeo q;
  int rc = make_geo(d, &g, &q);
  if (rc != 0) return rc;
  if ((rc = check_params(g, p)) != 0) return rc;
  if (packed == nullptr) return fail(VMLMF_E_BADARG, "null packed buffer");
  if (g.generic) return fail(VMLMF_E_UNSUPPORTED, "kept parameter images: not for the step-wise / clustered layers");
  const VPack P = vg_pack_layout(g, q.total);
  hipStream_t s = (hipStream_t)stream;
  float* img = (float*)packed + PK_HDR;
  g_packreg.put(packed, pack_signature(g, P, q));
  const RefP rp = to_refp(p);
  Scope sc(0, s);
  if ((rc = hip_fail(launch_pack(g, rp, P, img, s), "pack")) != 0) return rc;
  if (g.rb && (rc = hip_fail(launch_rb_pack(g, q, rp, img + P.RB, s), "rb_pack")) != 0) return rc;
  return 0;
}

int vmlmf_seq_forward(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                      const float* c0, float* y, float* hT, float* cT, void* reserve, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return vmlmf_seq_forward_packed(d, p, x, h0, c0, y, hT, cT, reserve, workspace, workspace_bytes, stream, nullptr);
}

int vmlmf_seq_forward_packed(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                             const float* c0, float* y, float* hT, float* cT, void* reserve, void* workspace,
                             size_t workspace_bytes, void* stream, const void* packed) {
  vmlmf_extra ex = {};   // every field the caller does not set is a null pointer (drop, ce, ...)
  ex.packed = packed, ex.head = nullptr, ex.ce = nullptr;
  return vmlmf_seq_forward_ex(d, p, x, h0, c0, y, hT, cT, reserve, workspace, workspace_bytes, stream, &ex);
}

static int check_head(const VGeo& g, const vmlmf_head* hd, bool fwd) {
  if (hd == nullptr || hd->classes == 0) return 0;
  if (hd->classes < 0 || hd->classes > head_max_classes()) return fail(VMLMF_E_UNSUPPORTED, "head: 1..32 classes");
  if (hd->weight == nullptr || (fwd ? hd->logits == nullptr : hd->dlogits == nullptr)) return fail(VMLMF_E_BADARG, "head: null pointer");
  return 0;
}

// vmlmf_dropout of a call -> kernel arguments (vmlmf_dropout.h); rc != 0: bad arguments / a layer whose kernels do not take it
static int make_drop(const vmlmf_dropout* dr, const VGeo& g, bool forward, DropArgs* out) {
  memset(out, 0, sizeof(*out));
  if (dr == nullptr) return 0;
  if (!(dr->p >= 0.f && dr->p < 1.f)) return fail(VMLMF_E_BADARG, "dropout: p must be in [0, 1)");
  if (dr->state == nullptr || (forward && dr->y_dropped == nullptr)) return fail(VMLMF_E_BADARG, "dropout: null state / y_dropped");
  if (!g.rb || g.syT != (long long)g.B * g.H)
    return fail(VMLMF_E_UNSUPPORTED, "dropout inside the layer's launches: row-block layers in the time-major layout (vmlmf_dropout_fused)");
  out->state = reinterpret_cast<const unsigned long long*>(dr->state), out->yd = forward ? dr->y_dropped : nullptr;
  out->thresh = drop_thresh(dr->p), out->scale = 1.f / (1.f - dr->p), out->site = dr->site;
  return 0;
}

int vmlmf_seq_forward_ex(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                         const float* c0, float* y, float* hT, float* cT, void* reserve, void* workspace,
                         size_t workspace_bytes, void* stream, const vmlmf_extra* ex) {
  const void* packed = ex != nullptr ? ex->packed : nullptr;
  const vmlmf_head* head = (ex != nullptr && ex->head != nullptr && ex->head->classes != 0) ? ex->head : nullptr;
  VGeo g;
  RbGeo q;
  int rc = take_status();
  if (rc != 0) return rc;
  if ((rc = make_geo(d, &g, &q)) != 0) return rc;
  if ((rc = check_params(g, p)) != 0) return rc;
  if (x == nullptr || y == nullptr || workspace == nullptr) return fail(VMLMF_E_BADARG, "null x / y / workspace");
  if (g.training && reserve == nullptr) return fail(VMLMF_E_BADARG, "training forward needs a reserve buffer");
  if ((rc = check_head(g, head, true)) != 0) return rc;
  // the classifier rides inside the VALU recurrent kernels; the other families run the stand-alone head kernel after
  // their recurrence (same values up to summation order)
  const bool head_inside = head != nullptr && !g.rb && !g.generic;
  if (head != nullptr && !head_inside && hT == nullptr) return fail(VMLMF_E_BADARG, "head on this layer needs the hT output");
  DropArgs drop;
  if ((rc = make_drop(ex != nullptr ? ex->drop : nullptr, g, true, &drop)) != 0) return rc;
  const vmlmf_ce* ce = ex != nullptr ? ex->ce : nullptr;
  if (ce != nullptr) {
    if (head == nullptr) return fail(VMLMF_E_BADARG, "ce: the criterion rides on the classifier's logits (extra.head)");
    if (!ce->target || !ce->loss || !ce->nvalid || !ce->lse || !ce->ticket) return fail(VMLMF_E_BADARG, "ce: null pointer");
  }
  // the criterion behind a classifier that is a launch of its own: a launch too (same values up to the mean's summation order)
  auto ce_after = [&]() -> int {
    if (ce == nullptr) return 0;
    Scope sc(10, (hipStream_t)stream);
    const hipError_t e = launch_ce_fwd(g.B, head->classes, head->logits, (const long long*)ce->target, (long long)ce->ignore_index, ce->loss,
                                       ce->lse, ce->nvalid, ce->dlogits_unit, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail((int)e, "ce_fwd");
  };
  const VPack P = vg_pack_layout(g, q.total);
  const Layout L = make_layout(g, P, q);
  if (workspace_bytes < (size_t)L.f_total * sizeof(float))
    return fail(VMLMF_E_WORKSPACE, "workspace smaller than vmlmf_query() reported");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* rs = (float*)reserve;
  float* pack = g.training ? rs + L.r_pack : ws + L.f_pack;
  if (packed != nullptr) {   // the caller's image (vmlmf_pack_params): nothing is packed here
    if (g.generic) return fail(VMLMF_E_UNSUPPORTED, "kept parameter images: not for the step-wise / clustered layers");
    if ((rc = check_packed(packed, g, P, q)) != 0) return rc;
    pack = const_cast<float*>((const float*)packed) + PK_HDR;
  }
  float* gx = ws + L.f_gx;
  const RefP rp = to_refp(p);
  const bool direct = packed == nullptr && direct_ok(g, p);
  if (packed == nullptr && !direct) {
    Scope sc(0, s);
    if ((rc = hip_fail(launch_pack(g, rp, P, pack, s), "pack")) != 0) return rc;
  }
  // narrow-input layers compute the x-projection inside rec_fwd_kernel (VMLMF_XWAVE=0: always the separate launch)
  const bool xwave = g_xwave && vg_xwave_ok(g);
  float* const qxbuf = g.training ? rs + L.r_qx : (g.generic ? ws + L.f_qx : nullptr);
  if (!xwave) {
    Scope sc(1, s);
    if ((rc = hip_fail(launch_xproj(g, P, pack, x, gx, qxbuf, s), "xproj")) != 0)
      return rc;
  }
  if (g.rb) {
    if (packed == nullptr) {
      Scope sc(0, s);
      if ((rc = hip_fail(launch_rb_pack(g, q, rp, pack + P.RB, s, reinterpret_cast<unsigned*>(ws + L.f_flag)), "rb_pack")) != 0) return rc;
    }
    RbIo io;
    memset(&io, 0, sizeof(io));
    io.flags_zeroed = packed == nullptr ? 1 : 0;
    io.gx = gx, io.EH = pack + P.EH, io.h0 = h0, io.c0 = c0, io.img = pack + P.RB, io.y = y, io.hT = hT, io.cT = cT;
    io.gates = g.training ? rs + L.r_gates : nullptr, io.cs = g.training ? rs + L.r_cs : nullptr;
    io.Qs = g.training ? rs + L.r_Qs : nullptr;
    io.xq = ws + L.f_xq, io.flag = reinterpret_cast<unsigned*>(ws + L.f_flag), io.status = status_word(s);
    io.drop = drop;
    {
      Scope sc(2, s);
      if ((rc = hip_fail(launch_rb_fwd(g, q, io, s), "rb_fwd")) != 0) return rc;
    }
    if (head != nullptr) {
      Scope sc(8, s);
      const hipError_t e = launch_head_fwd(g.B, g.H, head->classes, hT, g.H, head->weight, head->bias, head->logits, s);
      if (e != hipSuccess) return hip_fail((int)e, "head_fwd");
    }
    return ce_after();
  }
  if (g.generic) {
    GenericBuf w;
    memset(&w, 0, sizeof(w));
    w.gx = gx, w.EH = pack + P.EH, w.h0 = h0, w.c0 = c0, w.Ud = pack + P.UD, w.Vd = pack + P.VD;
    w.UdT = pack + P.UDT;   // the skinny products read both operands along k
    w.zeros = ws + L.f_zeros, w.y = y, w.hT = hT, w.cT = cT;
    w.gates = g.training ? rs + L.r_gates : nullptr, w.cs = g.training ? rs + L.r_cs : nullptr;
    w.Qs = g.training ? rs + L.r_Qs : nullptr, w.Qtmp = ws + L.f_Qtmp, w.P = ws + L.f_P, w.ccar = ws + L.f_ccar;
    w.part = ws + L.f_part, w.part_cap = (long long)VG_GEMM_SPLIT * ((g.B + 63) / 64 * 64) * ((g.G * g.KH + 63) / 64 * 64);
    w.ticket = reinterpret_cast<int*>(pack + P.TKT), w.ticket_cap = VG_GEMM_TICKETS;
    if (h0 == nullptr) {
      rc = (int)hipMemsetAsync(ws + L.f_zeros, 0, sizeof(float) * (size_t)g.B * g.H, s);
      if (rc != 0) return hip_fail(rc, "memset");
    }
    {
      Scope sc(2, s);
      if ((rc = hip_fail(generic_forward(g, w, s), "generic_forward")) != 0) return rc;
    }
    if (head != nullptr) {
      Scope sc(8, s);
      const hipError_t e = launch_head_fwd(g.B, g.H, head->classes, hT, g.H, head->weight, head->bias, head->logits, s);
      if (e != hipSuccess) return hip_fail((int)e, "head_fwd");
    }
    return ce_after();
  }
  FwdArgs a;
  a.gx = gx, a.VE = pack + P.VE, a.UR = pack + P.UR, a.EH = pack + P.EH, a.h0 = h0, a.c0 = c0;
  a.y = y, a.hT = hT, a.cT = cT, a.trash = ws + L.f_trash;
  a.gates = g.training ? rs + L.r_gates : nullptr;
  a.cs = g.training ? rs + L.r_cs : nullptr;
  a.Qs = g.training ? rs + L.r_Qs : nullptr;
  XwArgs xw;
  xw.x = x, xw.UXP = pack + P.UXP, xw.WXD = pack + P.WXD, xw.BBT = pack + P.BBT;
  memset(&xw.hd, 0, sizeof(xw.hd));
  if (head_inside) xw.hd.W = head->weight, xw.hd.bias = head->bias, xw.hd.logits = head->logits, xw.hd.C = head->classes;
  memset(&xw.ce, 0, sizeof(xw.ce));
  if (head_inside && ce != nullptr && g.R == 1 && g.B < 65536) {   // one batch row per workgroup: the row's terms are workgroup-local
    xw.ce.tgt = (const long long*)ce->target, xw.ce.ignore = (long long)ce->ignore_index, xw.ce.loss = ce->loss, xw.ce.nvalid = ce->nvalid;
    xw.ce.lse = ce->lse, xw.ce.dz = ce->dlogits_unit, xw.ce.ticket = (unsigned long long*)ce->ticket;
  }
  a.xwave = xwave ? 1 : 0, a.qxw = g.training ? rs + L.r_qx : nullptr;
  xw.BH = nullptr, xw.DX = nullptr, xw.direct = 0, xw.pad = 0;
  if (direct) {   // the reference's own tensors in the places of the images (vmlmf_direct.inc)
    a.VE = p->v_h[0], a.UR = p->u_h[0], a.EH = p->dia_h, a.xwave = 3;
    xw.UXP = p->u_x, xw.WXD = p->v_x, xw.BBT = p->b_x, xw.BH = p->b_h, xw.DX = p->dia_x, xw.direct = 1;
  }
  a.prog = g.training ? reinterpret_cast<unsigned*>(rs + L.r_prog) : nullptr;
  {
    Scope sc(2, s);
    if (xwave && ((g_rec3 & 1) || ((g_rec3 & 4) && g.nwg > device_cus())) && rec3_fwd_supported(g)) {
      if ((rc = hip_fail(launch_rec3_fwd(g, a, xw, s), "rec3_fwd")) != 0) return rc;
    } else if ((rc = hip_fail(launch_rec_fwd(g, a, xw, s), "rec_fwd")) != 0) return rc;
  }
  if (ce != nullptr && xw.ce.tgt == nullptr && (rc = ce_after()) != 0) return rc;   // (a batch beyond the ticket's 16-bit row count)
  return debug_status(s);
}

int vmlmf_seq_backward(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                       const float* c0, const float* y, const void* reserve, const float* dy,
                       const float* dhT, const float* dcT, float* dx, float* dh0, float* dc0,
                       const vmlmf_grads* gr, void* workspace, size_t workspace_bytes, void* stream) {
  return vmlmf_seq_backward_packed(d, p, x, h0, c0, y, reserve, dy, dhT, dcT, dx, dh0, dc0, gr, workspace, workspace_bytes, stream,
                                   nullptr);
}

int vmlmf_seq_backward_packed(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                              const float* c0, const float* y, const void* reserve, const float* dy,
                              const float* dhT, const float* dcT, float* dx, float* dh0, float* dc0,
                              const vmlmf_grads* gr, void* workspace, size_t workspace_bytes, void* stream,
                              const void* packed) {
  vmlmf_extra ex = {};   // every field the caller does not set is a null pointer (drop, ce, ...)
  ex.packed = packed, ex.head = nullptr, ex.ce = nullptr;
  return vmlmf_seq_backward_ex(d, p, x, h0, c0, y, reserve, dy, dhT, dcT, dx, dh0, dc0, gr, workspace, workspace_bytes, stream, &ex);
}

int vmlmf_seq_backward_ex(const vmlmf_desc* d, const vmlmf_params* p, const float* x, const float* h0,
                          const float* c0, const float* y, const void* reserve, const float* dy,
                          const float* dhT, const float* dcT, float* dx, float* dh0, float* dc0,
                          const vmlmf_grads* gr, void* workspace, size_t workspace_bytes, void* stream,
                          const vmlmf_extra* ex) {
  const void* packed = ex != nullptr ? ex->packed : nullptr;
  const vmlmf_head* head = (ex != nullptr && ex->head != nullptr && ex->head->classes != 0) ? ex->head : nullptr;
  VGeo g;
  RbGeo q;
  int rc = take_status();
  if (rc != 0) return rc;
  if ((rc = make_geo(d, &g, &q)) != 0) return rc;
  if ((rc = check_params(g, p)) != 0) return rc;
  if (x == nullptr || y == nullptr || reserve == nullptr || workspace == nullptr || gr == nullptr)
    return fail(VMLMF_E_BADARG, "null x / y / reserve / workspace / grads");
  if ((rc = check_pointers(g, gr, "grads")) != 0) return rc;
  if ((rc = check_head(g, head, false)) != 0) return rc;
  const bool head_inside = head != nullptr && !g.rb && !g.generic;
  DropArgs drop;
  if ((rc = make_drop(ex != nullptr ? ex->drop : nullptr, g, false, &drop)) != 0) return rc;
  const VPack P = vg_pack_layout(g, q.total);
  const Layout L = make_layout(g, P, q);
  if (workspace_bytes < (size_t)L.b_total * sizeof(float))
    return fail(VMLMF_E_WORKSPACE, "workspace smaller than vmlmf_query() reported");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const float* rs = (const float*)reserve;
  // final hidden state of the layer = last time slice of y
  const float* hlast = y + (size_t)(g.T - 1) * g.syT;
  HeadBwd hb;
  memset(&hb, 0, sizeof(hb));
  if (head != nullptr && !head_inside) {
    // stand-alone head kernel: dh into scratch, which then is the dhT of the recurrence
    if (dhT != nullptr) return fail(VMLMF_E_UNSUPPORTED, "head together with an explicit dhT: only on the VALU recurrent kernels");
    float* tmp = ws + L.b_headdh;
    Scope sc(9, s);
    const hipError_t e = launch_head_bwd(g.B, g.H, head->classes, hlast, g.syB, head->weight, head->dlogits, tmp, head->dweight,
                                         head->dbias, s);
    if (e != hipSuccess) return hip_fail((int)e, "head_bwd");
    dhT = tmp;
  } else if (head_inside) {
    hb.W = head->weight, hb.dl = head->dlogits, hb.hlast = hlast, hb.ldh = g.syB, hb.dW = head->dweight, hb.db = head->dbias;
    hb.C = head->classes;
  }
  WRide ride;
  memset(&ride, 0, sizeof(ride));
  bool inrow = false;
  const float* pack = rs + L.r_pack;
  if (packed != nullptr) {   // the image the matching forward was given
    if (g.generic) return fail(VMLMF_E_UNSUPPORTED, "kept parameter images: not for the step-wise / clustered layers");
    if ((rc = check_packed(packed, g, P, q)) != 0) return rc;
    pack = (const float*)packed + PK_HDR;
  }
  if (g.rb) {
    RbIo io;
    memset(&io, 0, sizeof(io));
    io.gates = const_cast<float*>(rs + L.r_gates), io.cs = const_cast<float*>(rs + L.r_cs), io.EH = pack + P.EH;
    io.img = pack + P.RB, io.dy = dy, io.dhT = dhT, io.dcT = dcT, io.dpre = ws + L.b_dpre, io.dQs = ws + L.b_dQs;
    io.dh0 = dh0, io.dc0 = dc0, io.xq = ws + L.b_xq, io.flag = reinterpret_cast<unsigned*>(ws + L.b_flag), io.status = status_word(s);
    io.drop = drop;
    {
      Scope sc(3, s);
      if ((rc = hip_fail(launch_rb_bwd(g, q, io, s), "rb_bwd")) != 0) return rc;
    }
    if (g.generic) {   // large layer: dqx as one skinny product over all rows, then dx
      GenericBuf w;
      memset(&w, 0, sizeof(w));
      w.dpre = ws + L.b_dpre, w.VxT = pack + P.VXTT, w.dqx = ws + L.b_dqx, w.dx = dx, w.UXP = pack + P.UXP, w.EXT = pack + P.EXT;
      w.part = ws + L.b_part, w.part_cap = (long long)VG_GEMM_SPLIT * ((g.B + 63) / 64 * 64) * ((g.G * g.KH + 63) / 64 * 64);
      w.ticket = reinterpret_cast<int*>(const_cast<float*>(pack + P.TKT)), w.ticket_cap = VG_GEMM_TICKETS;
      Scope sc(4, s);
      if ((rc = hip_fail(generic_dqx_dx(g, w, s), "dqx_dx")) != 0) return rc;
    } else {
      WgxArgs wx;
      wx.dpre = ws + L.b_dpre, wx.VRX = pack + P.VRX, wx.UXO = pack + P.UXO, wx.EXI = pack + P.EXI;
      wx.dx = dx, wx.dqx = ws + L.b_dqx;
      Scope sc(4, s);
      if ((rc = hip_fail(launch_wgrad_x(g, wx, s), "dqx_dx")) != 0) return rc;
    }
  } else if (g.generic) {
    GenericBuf w;
    memset(&w, 0, sizeof(w));
    w.EH = pack + P.EH, w.gates = const_cast<float*>(rs + L.r_gates), w.cs = const_cast<float*>(rs + L.r_cs);
    w.dy = dy, w.dhT = dhT, w.dcT = dcT, w.UdT = pack + P.UDT, w.VdT = pack + P.VDT, w.VxT = pack + P.VXTT;
    w.UXP = pack + P.UXP, w.EXT = pack + P.EXT, w.Vd = pack + P.VD;
    w.dpre = ws + L.b_dpre, w.dQs = ws + L.b_dQs, w.dHrec = ws + L.b_dHrec, w.ehterm = ws + L.b_ehterm;
    w.dcar = ws + L.b_dcar, w.dh0 = dh0, w.dc0 = dc0, w.dqx = ws + L.b_dqx, w.dx = dx;
    w.part = ws + L.b_part, w.part_cap = (long long)VG_GEMM_SPLIT * ((g.B + 63) / 64 * 64) * ((g.G * g.KH + 63) / 64 * 64);
    w.ticket = reinterpret_cast<int*>(const_cast<float*>(pack + P.TKT)), w.ticket_cap = VG_GEMM_TICKETS;
    {
      Scope sc(3, s);
      if ((rc = hip_fail(generic_backward(g, w, s), "generic_backward")) != 0) return rc;
    }
  } else {
  BwdArgs a;
  a.gates = rs + L.r_gates, a.cs = rs + L.r_cs, a.c0 = c0, a.dy = dy, a.dhT = dhT, a.dcT = dcT;
  a.VR = pack + P.VR, a.UE = pack + P.UE, a.EH = pack + P.EH, a.VE = pack + P.VE;
  a.dpre = ws + L.b_dpre, a.dQs = ws + L.b_dQs, a.dh0 = dh0, a.dc0 = dc0, a.trash = ws + L.b_trash;
  a.hd = hb;
  // weight gradients inside the rows' workgroups (vmlmf_rec4.inc)?  Layers it covers whose input needs no gradient; automatic:
  // batches beyond the riding workers' range (up to there the idle CUs form the products for free)
  inrow = g_inrow != 0 && dx == nullptr && rec4_bwd_supported(g) && (g_inrow > 0 || g.B > g_wride_maxb);
  if (inrow) {
    const WghArgs wh = wgrad_args(L, x, y, h0, rs, ws);
    ride.a.x = wh.x, ride.a.y = wh.y, ride.a.h0 = wh.h0, ride.a.Qs = wh.Qs, ride.a.P = wh.wpart;
  } else {
    plan_wride(g, L, x, y, h0, rs, ws, &ride, s, p->v_x);
    if (ride.K > 0 && !((g_rec3 & 2) && rec3_bwd_supported(g)) && !rec_bwd_rides(g)) memset(&ride, 0, sizeof(ride));
  }
  a.wr = ride;
  const bool direct = packed == nullptr && direct_ok(g, p);   // the forward of this call packed nothing
  if (direct) {
    // the backward builds its own images where its workgroups have a CU each (riding workers, weight gradients in the rows'
    // workgroups); elsewhere, and for the input's gradient (x-side images), this call packs after all
    const bool own = inrow || ride.K > 0;
    if (own) a.VE = p->v_h[0], a.UE = p->u_h[0], a.EH = p->dia_h, a.wr.direct = 1;
    if (!own || !(g.foldx && dx == nullptr)) {
      Scope sc(0, s);
      if ((rc = hip_fail(launch_pack(g, to_refp(p), P, const_cast<float*>(pack), s), "pack")) != 0) return rc;
    }
  }
  {
    Scope sc(3, s);
    if (inrow) {
      if ((rc = hip_fail(launch_rec4_bwd(g, a, s), "rec4_bwd")) != 0) return rc;
    } else if ((g_rec3 & 2) && rec3_bwd_supported(g)) {
      if ((rc = hip_fail(launch_rec3_bwd(g, a, s), "rec3_bwd")) != 0) return rc;
    } else if ((rc = hip_fail(launch_rec_bwd(g, a, s), "rec_bwd")) != 0) return rc;
  }
  WgxArgs wx;
  wx.dpre = ws + L.b_dpre, wx.VRX = pack + P.VRX, wx.UXO = pack + P.UXO, wx.EXI = pack + P.EXI;
  wx.dx = dx, wx.dqx = ws + L.b_dqx;
  if (!(g.foldx && dx == nullptr)) {   // with the x-fold dqx only feeds dx
    Scope sc(4, s);
    if ((rc = hip_fail(launch_wgrad_x(g, wx, s), "dqx_dx")) != 0) return rc;
  }
  }  // persistent path
  if ((rc = backward_tail(g, L, p, gr, x, y, h0, rs, ws, hb, s, &ride, inrow ? g.B : 0)) != 0) return rc;
  return debug_status(s);
}


// ---- stacked layers: wavefront launches (vmlmf_wave.inc) ----
namespace {
// VMLMF_WF_BWD=0: the stack's backward chains the per-layer kernels (A/B runs and bring-up); the forward is the wavefront
// launch either way
const bool g_wf_bwd = env_int("VMLMF_WF_BWD", 1) != 0;
// VMLMF_PACK_SLIM=0: the stack's pack launch produces every image of pack_kernel (A/B; the chained backward needs them anyway)
const bool g_pack_slim = env_int("VMLMF_PACK_SLIM", 1) != 0;
// VMLMF_FINISH_UNITS=0: behind a stack's weight-gradient launch, reduce_cg_stack_kernel + finish_stack_kernel instead of the one
// finishing launch (A/B)
const bool g_finish_units = env_int("VMLMF_FINISH_UNITS", 1) != 0;

// (No lambdas in the initialisers of this block: this unnamed namespace is reopened INSIDE the file's extern "C" block, and hipcc numbers
//  the lambdas of a namespace per enclosing linkage specification - "(anonymous namespace)::{lambda()#2}" here got the same mangled name
//  as lambda #2 of the first block at the top of the file, and ONE body served both: the switches below came up with another switch's
//  default whatever the environment said.  Found in round 6 through a run-time flag that read 1 with nothing set; VMLMF_WF_BWD had been
//  answered by VMLMF_RB's lambda since round 2.)
// VMLMF_RBX=0 / vmlmf_tune("rbx", 0): clustered layers are never stacked into one launch (the caller chains them; A/B runs)
int g_rbx = env_int("VMLMF_RBX", 1);
// the stacks' finishing launch sums the partial blocks itself (no reduce launch): 0 = never (default: measured slower - config C's
// finish_stack_kernel 25.5 us against reduce 8.1 + finish 6.2, the repeated block sums of the d(ex) / d(eh) rows; two PTB group
// layers at 32 rows 0.710 ms with, 0.702 without), -1 = the clustered stacks, 1 = the wavefront stacks too.  VMLMF_FFB /
// vmlmf_tune("ffb"); parity-tested both ways
int g_ffb = env_int("VMLMF_FFB", 0);

struct StackPlan {
  int L;
  bool rbx;    // the clustered form (vmlmf_rbx.hip): every layer on clusters of workgroups, all layers in one launch per direction
  RbGeo q;     // ... its geometry (all layers alike)
  VGeo g[WF_MAXL];
  VPack P[WF_MAXL];
  WfPack W;
  Layout lay[WF_MAXL];
  long long ws_flag, ws_layer[WF_MAXL], ws_dx[WF_MAXL], ws_total;   // float offsets in the workspace
  long long flag_words;
};

static int stack_plan(int L, const vmlmf_stack_layer* ly, StackPlan* out) {
  if (ly == nullptr) return fail(VMLMF_E_BADARG, "stack: null layers");
  if (L < 1 || L > WF_MAXL) return fail(VMLMF_E_UNSUPPORTED, "stack: 1..4 layers");
  StackPlan& S = *out;
  S.L = L;
  S.rbx = false;
  {   // layers on clusters of workgroups (factors beyond one CU): the clustered form, or nothing
    RbGeo q0;
    vmlmf_desc d0 = ly[0].desc;
    VGeo g0;
    if (d0.dtype == VMLMF_DT_F32 && make_geo(&d0, &g0, &q0) == 0 && g0.generic && g0.rb > 1) {
      if (!g_rbx) return fail(VMLMF_E_UNSUPPORTED, "stack: the clustered form is switched off (VMLMF_RBX=0)");
      if (L < 2 && g_rbx != 2) return fail(VMLMF_E_UNSUPPORTED, "stack: a single clustered layer runs as vmlmf_seq_forward");
      if (L > RBX_MAXL) return fail(VMLMF_E_UNSUPPORTED, "stack: at most four clustered layers");
      for (int l = 0; l < L; ++l) {
        RbGeo ql;
        vmlmf_desc dd = ly[l].desc;
        const int rc = make_geo(&dd, &S.g[l], &ql);
        if (rc != 0) return rc;
        const VGeo& g = S.g[l];
        if (g.variant != g0.variant || g.B != g0.B || g.T != g0.T || g.H != g0.H || g.I != g0.I || g.rw != g0.rw || g.ru0 != g0.ru0 || g.ru1 != g0.ru1 || g.G != g0.G ||
            g.time_major != g0.time_major || g.training != g0.training || dd.dtype != VMLMF_DT_F32 || g.rb != g0.rb)
          return fail(VMLMF_E_UNSUPPORTED, "stack: layers must agree in variant, B, T, sizes, ranks, layout and training flag");
        if (g.I != g.H || !g.time_major || g.sxT != g.syT || g.sxB != g.syB)
          return fail(VMLMF_E_UNSUPPORTED, "stack (clustered form): time-major layers with input_size == hidden_size");
      }
      // live rows per workgroup: the fewest (4, 8, 16) with which the clusters of ALL layers are co-resident, one workgroup per CU
      const int cus = device_cus();
      bool found = false;
      for (int rows = 4; rows <= 16 && !found; rows *= 2) {
        RbGeo q;
        if (!rb_geometry(g0, g0.rb, &q, rows, 1) || !rbx_supported(g0, q)) continue;
        if ((long long)L * q.nrb * q.S > cus) continue;
        S.q = q, found = true;
      }
      if (!found)
        return fail(VMLMF_E_UNSUPPORTED, "stack (clustered form): V3 / V4 layers with w_rank 17..32 whose clusters are co-resident for all layers "
                                         "(L x ceil(B / 16) x 16 workgroups <= CUs)");
      S.rbx = true;
      S.flag_words = 0;
      memset(&S.W, 0, sizeof(S.W));
      long long o = 0;
      S.ws_flag = 0;
      for (int l = 0; l < L; ++l) {
        S.P[l] = vg_pack_layout(S.g[l], S.q.total, 0);
        S.lay[l] = make_layout(S.g[l], S.P[l], S.q);
        const long long per = S.lay[l].f_total > S.lay[l].b_total ? S.lay[l].f_total : S.lay[l].b_total;
        S.ws_layer[l] = o, o += align64(per);
        S.ws_dx[l] = o, o += align64(l > 0 ? (long long)g0.T * g0.B * g0.H : 0);
      }
      S.ws_total = o;
      return 0;
    }
  }
  // layers of a stack may differ in hidden_size (MyLSTM builds any hidden_layer_sizes, vmlmf.py:283-292; a VMLMF cell needs input_size
  // <= hidden_size, vmlmf.py:94, so the sizes cannot shrink): every layer then runs on the widest layer's wave count
  int Wmax = 0;
  for (int l = 0; l < L; ++l) {
    RbGeo q;
    VGeo gl;
    vmlmf_desc dd = ly[l].desc;
    if (dd.dtype == VMLMF_DT_BF16) dd.dtype = VMLMF_DT_F32;
    const int rc = make_geo(&dd, &gl, &q);
    if (rc != 0) return rc;
    Wmax = gl.W > Wmax ? gl.W : Wmax;
  }
  for (int l = 0; l < L; ++l) {
    RbGeo q;
    // dtype bf16 on a stack: below the batch where the bf16-MFMA row blocks pay (4096 rows: DESIGN.md section 4.8) the wavefront
    // kernels run it with fp32 arithmetic and a bf16 GATE TAPE (VGeo::bt: half the tape bytes written and read back; one-group
    // layers of padded rank 16 / 24); from there on, and for the layers those instantiations do not cover, the caller chains the
    // row-block kernels (VMLMF_E_UNSUPPORTED here)
    vmlmf_desc dd = ly[l].desc;
    const bool bt = dd.dtype == VMLMF_DT_BF16;
    if (bt) {
      if (dd.B >= 4096) return fail(VMLMF_E_UNSUPPORTED, "stack: dtype bf16 at 4096 rows and more runs the row-block bf16-MFMA kernels layer by layer");
      dd.dtype = VMLMF_DT_F32;
    }
    const int rc = make_geo(&dd, &S.g[l], &q, Wmax);
    if (rc != 0) return rc;
    if (bt) {
      const int K = wf_width(S.g[l]);
      if (S.g[l].G != 1 || !(K == 16 || K == 24) || !g_wf_bwd)
        return fail(VMLMF_E_UNSUPPORTED, "stack: the bf16 gate tape covers one-group layers of padded rank 16 / 24");
      S.g[l].bt = 1;
    }
    const VGeo& g = S.g[l];
    if (!wf_supported(g))
      return fail(VMLMF_E_UNSUPPORTED, "stack: layer not covered by the wavefront kernels (V1-V3, V5, V6; at most four waves of hidden units; padded ranks "
                                       "<= 24, or 32 with at most three waves; fp32)");
    if (l > 0) {
      const VGeo& g0 = S.g[0];
      if (g.variant != g0.variant || g.B != g0.B || g.T != g0.T || g.KH != g0.KH || g.KX != g0.KX || g.ru0 != g0.ru0 || g.ru1 != g0.ru1 || g.G != g0.G || g.rw != g0.rw || g.bt != g0.bt ||
          g.time_major != g0.time_major || g.training != g0.training || g.W != g0.W)
        return fail(VMLMF_E_UNSUPPORTED, "stack: layers must agree in variant, B, T, ranks, layout and training flag");
      if (g.I != S.g[l - 1].H) return fail(VMLMF_E_SHAPE, "stack: layer l > 0 reads the layer below: its input_size must equal that layer's hidden_size");
      if (g.H != g0.H && (g.bt || g.G != 1 || g.KH != g.KX))
        return fail(VMLMF_E_UNSUPPORTED, "stack: layers of different hidden sizes: one-group layers, fp32 tapes, equal padded ranks on both sides");
    }
  }
  {   // the batched weight-gradient launch of these stacks holds one workgroup per CU: few enough chunks for one round (vmlmf_wgrad4.hip)
    const int rc2 = wgrad4_chunk_rows(L, S.g, device_cus());
    if (rc2 > 0)
      for (int l = 0; l < L; ++l) {
        const int TB = S.g[l].T * S.g[l].B;
        S.g[l].RC2 = rc2, S.g[l].nchunk = (TB + rc2 - 1) / rc2;
      }
  }
  S.W = wf_pack_layout(S.g[0]);
  long long o = 0;
  S.flag_words = ((long long)(L > 1 ? L - 1 : 0) * S.g[0].B + 1) * WF_FLAG_STRIDE;
  S.ws_flag = o, o += align64(S.flag_words);
  RbGeo q0;
  memset(&q0, 0, sizeof(q0));
  for (int l = 0; l < L; ++l) {
    S.P[l] = vg_pack_layout(S.g[l], 0, S.W.total);
    S.lay[l] = make_layout(S.g[l], S.P[l], q0);
    const long long per = S.lay[l].f_total > S.lay[l].b_total ? S.lay[l].f_total : S.lay[l].b_total;
    S.ws_layer[l] = o, o += align64(per);
    S.ws_dx[l] = o, o += align64(l > 0 ? (long long)S.g[0].T * S.g[0].B * S.g[l].I : 0);   // dx of layer l = dy of layer l - 1
  }
  S.ws_total = o;
  return 0;
}
// ---- the clustered form (vmlmf_rbx.hip)
static int rbx_drop(const vmlmf_dropout* dr, bool forward, DropArgs* out) {
  memset(out, 0, sizeof(*out));
  if (dr == nullptr) return 0;
  if (!(dr->p >= 0.f && dr->p < 1.f)) return fail(VMLMF_E_BADARG, "dropout: p must be in [0, 1)");
  if (dr->state == nullptr || (forward && dr->y_dropped == nullptr)) return fail(VMLMF_E_BADARG, "dropout: null state / y_dropped");
  out->state = reinterpret_cast<const unsigned long long*>(dr->state), out->yd = forward ? dr->y_dropped : nullptr;
  out->thresh = drop_thresh(dr->p), out->scale = 1.f / (1.f - dr->p), out->site = dr->site;
  return 0;
}

static int rbx_stack_forward(const StackPlan& S, const vmlmf_stack_layer* ly, const float* x, float* ws, hipStream_t s) {
  const int L = S.L;
  const bool training = S.g[0].training != 0;
  int rc;
  RbxFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.status = status_word(s), a.L = L;
  RefP rps[RBX_MAXL];
  float* packs[RBX_MAXL];
  float* imgs[RBX_MAXL];
  unsigned* fflags[RBX_MAXL];
  for (int l = 0; l < L; ++l) {
    const VGeo& g = S.g[l];
    if ((rc = check_params(g, ly[l].params)) != 0) return rc;
    if (ly[l].y == nullptr) return fail(VMLMF_E_BADARG, "stack: null y");
    if (training && ly[l].reserve == nullptr) return fail(VMLMF_E_BADARG, "stack: training forward needs the layers' reserve buffers");
    float* rs = (float*)ly[l].reserve;
    const Layout& Lr = S.lay[l];
    float* wl = ws + S.ws_layer[l];
    float* pack = training ? rs + Lr.r_pack : wl + Lr.f_pack;
    rps[l] = to_refp(ly[l].params), packs[l] = pack, imgs[l] = pack + S.P[l].RB, fflags[l] = reinterpret_cast<unsigned*>(wl + Lr.f_flag);
    RbxLayerF& w = a.l[l];
    if ((rc = rbx_drop(ly[l].drop, true, &w.drop)) != 0) return rc;
    // the layer's input: x, or the rows of the layer below (their dropped copy under dropout)
    w.x = l == 0 ? x : (ly[l - 1].drop != nullptr ? ly[l - 1].drop->y_dropped : ly[l - 1].y);
    w.EH = pack + S.P[l].EH, w.EXT = pack + S.P[l].EXT, w.BBT = pack + S.P[l].BBT, w.img = pack + S.P[l].RB;
    w.h0 = ly[l].h0, w.c0 = ly[l].c0, w.y = ly[l].y, w.hT = ly[l].hT, w.cT = ly[l].cT;
    w.gates = training ? rs + Lr.r_gates : nullptr, w.cs = training ? rs + Lr.r_cs : nullptr;
    w.Qs = training ? rs + Lr.r_Qs : nullptr, w.qx = training ? rs + Lr.r_qx : nullptr;
    w.xq = wl + Lr.f_xq, w.flag = reinterpret_cast<unsigned*>(wl + Lr.f_flag);
    w.pflag = l > 0 ? reinterpret_cast<unsigned*>(ws + S.ws_layer[l - 1] + S.lay[l - 1].f_flag) : nullptr;
    w.pub = l < L - 1 ? 1 : 0;
  }
  {   // every layer's parameter images in two launches (pack_kernel's for all layers, the clusters' MFMA operand images for all
      // layers; the second also clears the forward launch's epoch words)
    Scope sc(0, s);
    WfPack W0;
    memset(&W0, 0, sizeof(W0));
    if ((rc = hip_fail(launch_pack_stack(L, S.g, rps, S.P, W0, packs, nullptr, 0, nullptr, 0, s, PACK_CLUSTERED), "pack")) != 0) return rc;
    if ((rc = hip_fail(launch_rb_pack_stack(S.g[0], S.q, L, rps, imgs, fflags, s), "rb_pack")) != 0) return rc;
  }
  Scope sc(2, s);
  return hip_fail(launch_rbx_fwd(S.g[0], S.q, a, s), "rbx_fwd");
}

static int rbx_stack_backward(const StackPlan& S, const vmlmf_stack_layer* ly, const float* x, const float* dy, float* dx, float* ws,
                              hipStream_t s) {
  const int L = S.L;
  int rc;
  for (int l = 0; l < L; ++l) {
    if ((rc = check_params(S.g[l], ly[l].params)) != 0) return rc;
    if ((rc = check_pointers(S.g[l], ly[l].grads, "grads")) != 0) return rc;
    if (ly[l].y == nullptr || ly[l].reserve == nullptr) return fail(VMLMF_E_BADARG, "stack: null y / reserve");
  }
  RbxBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.status = status_word(s), a.L = L;
  float* dpres[RBX_MAXL];
  unsigned* flags[RBX_MAXL];
  for (int l = 0; l < L; ++l) {
    const Layout& Lr = S.lay[l];
    const float* rs = (const float*)ly[l].reserve;
    const float* pack = rs + Lr.r_pack;
    float* wl = ws + S.ws_layer[l];
    RbxLayerB& w = a.l[L - 1 - l];   // launch position 0 is the top layer: the producer comes first in the grid
    if ((rc = rbx_drop(ly[l].drop, false, &w.drop)) != 0) return rc;
    w.gates = rs + Lr.r_gates, w.cs = rs + Lr.r_cs, w.EH = pack + S.P[l].EH, w.EXT = pack + S.P[l].EXT, w.img = pack + S.P[l].RB;
    w.dy = l == L - 1 ? dy : ws + S.ws_dx[l + 1];
    w.dhT = ly[l].dhT, w.dcT = ly[l].dcT, w.dh0 = ly[l].dh0, w.dc0 = ly[l].dc0;
    w.dpre = wl + Lr.b_dpre, w.dQs = wl + Lr.b_dQs, w.dqx = wl + Lr.b_dqx;
    w.dx = l == 0 ? dx : ws + S.ws_dx[l];
    w.xq = wl + Lr.b_xq, w.flag = reinterpret_cast<unsigned*>(wl + Lr.b_flag);
    w.pflag = l < L - 1 ? reinterpret_cast<unsigned*>(ws + S.ws_layer[l + 1] + S.lay[l + 1].b_flag) : nullptr;
    w.pub = l > 0 ? 1 : 0;
    dpres[l] = wl + Lr.b_dpre, flags[l] = reinterpret_cast<unsigned*>(wl + Lr.b_flag);
  }
  {
    Scope sc(3, s);
    if ((rc = hip_fail(launch_rbx_zero(S.g[0], S.q, L, dpres, flags, s), "rbx_zero")) != 0) return rc;
    if ((rc = hip_fail(launch_rbx_bwd(S.g[0], S.q, a, s), "rbx_bwd")) != 0) return rc;
  }
  HeadBwd hb;
  memset(&hb, 0, sizeof(hb));
  // the batched half of every layer: weight-gradient products (per layer: the ring kernel fills the chip), then ONE launch that sums
  // every layer's partial blocks and ONE that writes every layer's reference-layout gradients
  const bool ring = g_wring != 0 && wgrad_ring_ok(S.g[0]) && (g_wring > 0 || (long long)S.g[0].T * S.g[0].B >= 1024);
  if (!ring) {
    for (int l = L - 1; l >= 0; --l) {
      const float* xl = l == 0 ? x : (ly[l - 1].drop != nullptr ? ly[l - 1].drop->y_dropped : ly[l - 1].y);
      if ((rc = backward_tail(S.g[l], S.lay[l], ly[l].params, ly[l].grads, xl, ly[l].y, ly[l].h0, (const float*)ly[l].reserve,
                              ws + S.ws_layer[l], hb, s)) != 0)
        return rc;
    }
    return 0;
  }
  ReduceCounts wcs[RBX_MAXL];
  const float* wparts[RBX_MAXL];
  float* cgs[RBX_MAXL];
  const float* ccgs[RBX_MAXL];
  RefP rps[RBX_MAXL];
  RefG ogs[RBX_MAXL];
  for (int l = L - 1; l >= 0; --l) {
    const float* xl = l == 0 ? x : (ly[l - 1].drop != nullptr ? ly[l - 1].drop->y_dropped : ly[l - 1].y);
    float* wl = ws + S.ws_layer[l];
    const WghArgs wh = wgrad_args(S.lay[l], xl, ly[l].y, ly[l].h0, (const float*)ly[l].reserve, wl);
    int nc[3] = {0, 0, 0};
    {
      Scope sc(5, s);
      const int rr = launch_wgrad_ring(S.g[l], wh, device_cus(), nc, s);
      if (rr == -3) {   // no LDS / instantiation for the ring on this device: the per-layer path for every layer from here
        for (int k = l; k >= 0; --k) {
          const float* xk = k == 0 ? x : (ly[k - 1].drop != nullptr ? ly[k - 1].drop->y_dropped : ly[k - 1].y);
          if ((rc = backward_tail(S.g[k], S.lay[k], ly[k].params, ly[k].grads, xk, ly[k].y, ly[k].h0, (const float*)ly[k].reserve,
                                  ws + S.ws_layer[k], hb, s)) != 0)
            return rc;
        }
        // (the layers above l: their blocks are formed, finish them one by one)
        for (int k = L - 1; k > l; --k) {
          {
            Scope sc6(6, s);
            if ((rc = hip_fail(launch_reduce(S.g[k], wparts[k], cgs[k], nullptr, s, wcs[k]), "reduce")) != 0) return rc;
          }
          Scope sc7(7, s);
          if ((rc = hip_fail(launch_finish(S.g[k], rps[k], cgs[k], ogs[k], hb, s, health_word(s)), "finish")) != 0) return rc;
        }
        return 0;
      }
      if ((rc = hip_fail(rr, "wgrad")) != 0) return rc;
    }
    wcs[l] = ReduceCounts{{nc[0], nc[1], nc[2]}};
    wparts[l] = wl + S.lay[l].b_wpart, cgs[l] = wl + S.lay[l].b_cgrad, ccgs[l] = cgs[l];
    rps[l] = to_refp(ly[l].params);
    const vmlmf_grads* gr = ly[l].grads;
    RefG& og = ogs[l];
    og.dia_x = gr->dia_x, og.dia_h = gr->dia_h, og.u_x = gr->u_x, og.v_x = gr->v_x, og.b_x = gr->b_x;
    og.b_h = gr->b_h, og.u_h0 = gr->u_h[0], og.u_h1 = gr->u_h[1], og.v_h0 = gr->v_h[0], og.v_h1 = gr->v_h[1];
    for (int k = 0; k < 4; ++k) og.wg[k] = gr->w_gate[k], og.ug[k] = gr->u_gate[k], og.bg[k] = gr->b_gate[k];
  }
  {   // one finishing launch where it covers the layers (one-group layers: the plain rank-32 PTB layers)
    bool fu = g_finish_units;
    for (int l = 0; l < L; ++l) fu = fu && finish_units_ok(S.g[l]);
    if (fu) {
      Scope sc(7, s);
      return hip_fail(launch_finish_units_stack(L, S.g, rps, ogs, hb, s, health_word(s), wparts, wcs), "finish");
    }
  }
  if (g_ffb != 0 && finish_from_blocks_ok(S.g[0])) {   // the finishing launch sums the (few) partial blocks itself
    Scope sc(7, s);
    return hip_fail(launch_finish_stack(L, S.g, rps, ccgs, ogs, hb, s, health_word(s), wparts, wcs), "finish");
  }
  {
    Scope sc(6, s);
    if ((rc = hip_fail(launch_reduce_stack(L, S.g, wparts, cgs, s, wcs), "reduce")) != 0) return rc;
  }
  {
    Scope sc(7, s);
    if ((rc = hip_fail(launch_finish_stack(L, S.g, rps, ccgs, ogs, hb, s, health_word(s)), "finish")) != 0) return rc;
  }
  return 0;
}
}  // namespace

int vmlmf_stack_dropout_fused(int L, const vmlmf_stack_layer* layers) {
  StackPlan S;
  if (stack_plan(L, layers, &S) != 0) return 0;
  return (S.rbx || (S.g[0].G == 1 && g_wf_bwd)) ? 1 : 0;   // the clustered form; the wavefront launches for one-group layers
}

int vmlmf_stack_query(int L, const vmlmf_stack_layer* layers, size_t* reserve_bytes, size_t* workspace_bytes) {
  StackPlan S;
  const int rc = stack_plan(L, layers, &S);
  if (rc != 0) return rc;
  for (int l = 0; l < L; ++l)   // (layer 0's reserve ends with the progress words of the backward launch: the forward clears them)
    if (reserve_bytes != nullptr) reserve_bytes[l] = (size_t)(S.lay[l].r_total + (l == 0 ? align64(S.flag_words) : 0)) * sizeof(float);
  if (workspace_bytes != nullptr) *workspace_bytes = (size_t)S.ws_total * sizeof(float);
  return 0;
}

int vmlmf_stack_forward(int L, const vmlmf_stack_layer* ly, const float* x, const vmlmf_head* head_in, void* workspace,
                        size_t workspace_bytes, void* stream) {
  StackPlan S;
  int rc = take_status();
  if (rc != 0) return rc;
  if ((rc = stack_plan(L, ly, &S)) != 0) return rc;
  if (x == nullptr || workspace == nullptr) return fail(VMLMF_E_BADARG, "stack: null x / workspace");
  const vmlmf_head* head = (head_in != nullptr && head_in->classes != 0) ? head_in : nullptr;
  if ((rc = check_head(S.g[L - 1], head, true)) != 0) return rc;
  if (workspace_bytes < (size_t)S.ws_total * sizeof(float)) return fail(VMLMF_E_WORKSPACE, "stack: workspace smaller than vmlmf_stack_query() reported");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const bool training = S.g[0].training != 0;
  if (S.rbx) {
    if (head != nullptr) return fail(VMLMF_E_UNSUPPORTED, "stack (clustered form): no classifier head");
    return rbx_stack_forward(S, ly, x, ws, s);
  }
  for (int l = 0; l < L; ++l)
    if (ly[l].drop != nullptr && S.g[0].G != 1)
      return fail(VMLMF_E_UNSUPPORTED, "stack: dropout inside the wavefront launches for one-group layers (vmlmf_stack_dropout_fused)");
  WfFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.c.flag = reinterpret_cast<unsigned*>(ws + S.ws_flag), a.c.L = L, a.c.status = status_word(s);
  if (head != nullptr) a.hd.W = head->weight, a.hd.bias = head->bias, a.hd.logits = head->logits, a.hd.C = head->classes;
  RefP rps[WF_MAXL];
  float* packs[WF_MAXL];
  for (int l = 0; l < L; ++l) {
    const VGeo& g = S.g[l];
    if ((rc = check_params(g, ly[l].params)) != 0) return rc;
    if (ly[l].y == nullptr) return fail(VMLMF_E_BADARG, "stack: null y");
    if (training && ly[l].reserve == nullptr) return fail(VMLMF_E_BADARG, "stack: training forward needs the layers' reserve buffers");
    float* rs = (float*)ly[l].reserve;
    const Layout& Lr = S.lay[l];
    float* pack = training ? rs + Lr.r_pack : ws + S.ws_layer[l] + Lr.f_pack;
    rps[l] = to_refp(ly[l].params), packs[l] = pack;
    WfFwdLayer& w = a.l[l];
    // the layer's input: x, or the rows of the layer below - their dropped copy under dropout (vmlmf_lm.py:438-439)
    w.x = l == 0 ? x : (ly[l - 1].drop != nullptr ? ly[l - 1].drop->y_dropped : ly[l - 1].y);
    if ((rc = rbx_drop(ly[l].drop, true, &a.drop[l])) != 0) return rc;
    w.sxT = g.sxT, w.sxB = g.sxB, w.I = g.I;
    w.syT = g.syT, w.syB = g.syB, w.H = g.H, w.Hg = g.Hg;
    const bool mixed = g.KH != g.KX;   // both sides at the wider padded rank: re-laid images in the WF region
    const float* wf = pack + S.P[l].WF;
    w.VE = mixed ? wf + S.W.VE : pack + S.P[l].VE, w.VXT = mixed ? wf + S.W.VXK : pack + S.P[l].VXT;
    w.EH = pack + S.P[l].EH, w.EXT = pack + S.P[l].EXT, w.BBT = pack + S.P[l].BBT;
    w.UR = wf + S.W.UR, w.URX = wf + S.W.URX;
    w.h0 = ly[l].h0, w.c0 = ly[l].c0, w.y = ly[l].y, w.hT = ly[l].hT, w.cT = ly[l].cT;
    w.gates = training ? rs + Lr.r_gates : nullptr, w.cs = training ? rs + Lr.r_cs : nullptr;
    w.Qs = training ? rs + Lr.r_Qs : nullptr, w.qx = training ? rs + Lr.r_qx : nullptr;
  }
  {
    // one launch: every layer's parameter images, and the progress words of this launch and of the backward one cleared
    unsigned* z0 = L > 1 ? reinterpret_cast<unsigned*>(ws + S.ws_flag) : nullptr;
    unsigned* z1 = (L > 1 && training) ? reinterpret_cast<unsigned*>((float*)ly[0].reserve + S.lay[0].r_total) : nullptr;
    Scope sc(0, s);
    if ((rc = hip_fail(launch_pack_stack(L, S.g, rps, S.P, S.W, packs, z0, (int)S.flag_words, z1, (int)S.flag_words, s,
                                            (g_wf_bwd && g_pack_slim) ? PACK_WAVEFRONT : PACK_ALL), "pack")) != 0) return rc;
  }
  {
    Scope sc(2, s);
    if ((rc = hip_fail(launch_wf_fwd(S.g[0], a, s), "wf_fwd")) != 0) return rc;
  }
  return 0;
}

int vmlmf_stack_backward(int L, const vmlmf_stack_layer* ly, const float* x, const float* dy, float* dx,
                         const vmlmf_head* head_in, void* workspace, size_t workspace_bytes, void* stream) {
  StackPlan S;
  int rc = take_status();
  if (rc != 0) return rc;
  if ((rc = stack_plan(L, ly, &S)) != 0) return rc;
  if (x == nullptr || workspace == nullptr) return fail(VMLMF_E_BADARG, "stack: null x / workspace");
  const vmlmf_head* head = (head_in != nullptr && head_in->classes != 0) ? head_in : nullptr;
  if ((rc = check_head(S.g[L - 1], head, false)) != 0) return rc;
  if (workspace_bytes < (size_t)S.ws_total * sizeof(float)) return fail(VMLMF_E_WORKSPACE, "stack: workspace smaller than vmlmf_stack_query() reported");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  for (int l = 0; l < L; ++l) {
    if ((rc = check_params(S.g[l], ly[l].params)) != 0) return rc;
    if ((rc = check_pointers(S.g[l], ly[l].grads, "grads")) != 0) return rc;
    if (ly[l].y == nullptr || ly[l].reserve == nullptr) return fail(VMLMF_E_BADARG, "stack: null y / reserve");
  }
  HeadBwd hb;      // per-layer kernels of the chained form: no classifier riding
  memset(&hb, 0, sizeof(hb));
  HeadBwd hb_top;  // the classifier on the top layer
  memset(&hb_top, 0, sizeof(hb_top));
  if (head != nullptr) {
    const VGeo& gt = S.g[L - 1];
    hb_top.W = head->weight, hb_top.dl = head->dlogits, hb_top.hlast = ly[L - 1].y + (size_t)(gt.T - 1) * gt.syT, hb_top.ldh = gt.syB;
    hb_top.dW = head->dweight, hb_top.db = head->dbias, hb_top.C = head->classes;
  }
  if (S.rbx) {
    if (head != nullptr) return fail(VMLMF_E_UNSUPPORTED, "stack (clustered form): no classifier head");
    return rbx_stack_backward(S, ly, x, dy, dx, ws, s);
  }
  const bool wave = g_wf_bwd;
  for (int l = 0; l < L; ++l)
    if (!wave && ly[l].drop != nullptr) return fail(VMLMF_E_UNSUPPORTED, "stack: dropout rides on the wavefront backward only (VMLMF_WF_BWD=0 is an A/B switch)");
  if (!wave && head != nullptr) return fail(VMLMF_E_UNSUPPORTED, "stack: the classifier rides on the wavefront backward only (VMLMF_WF_BWD=0 is an A/B switch)");
  if (wave) {
    WfBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.c.flag = reinterpret_cast<unsigned*>((float*)ly[0].reserve + S.lay[0].r_total), a.c.L = L, a.c.status = status_word(s);   // cleared by the forward
    a.hd = hb_top;
    for (int l = 0; l < L; ++l) {
      const VGeo& g = S.g[l];
      const Layout& Lr = S.lay[l];
      const float* rs = (const float*)ly[l].reserve;
      const float* pack = rs + Lr.r_pack;
      float* wl = ws + S.ws_layer[l];
      WfBwdLayer& w = a.l[L - 1 - l];   // launch position 0 is the top layer
      if ((rc = rbx_drop(ly[l].drop, false, &a.drop[L - 1 - l])) != 0) return rc;
      if (ly[l].drop != nullptr && S.g[0].G != 1) return fail(VMLMF_E_UNSUPPORTED, "stack: dropout inside the wavefront launches for one-group layers");
      w.gates = rs + Lr.r_gates, w.cs = rs + Lr.r_cs;
      w.dy = l == L - 1 ? dy : ws + S.ws_dx[l + 1];
      w.dhT = ly[l].dhT, w.dcT = ly[l].dcT, w.dh0 = ly[l].dh0, w.dc0 = ly[l].dc0;
      const bool mixed = g.KH != g.KX;
      const float* wf = pack + S.P[l].WF;
      w.UE = mixed ? wf + S.W.UE : pack + S.P[l].UE, w.UXO = mixed ? wf + S.W.UXK : pack + S.P[l].UXO;
      w.EH = pack + S.P[l].EH, w.EXI = pack + S.P[l].EXI;
      w.VR = wf + S.W.VR, w.VRX = wf + S.W.VRX;
      w.dpre = wl + Lr.b_dpre, w.dQs = wl + Lr.b_dQs, w.dqx = wl + Lr.b_dqx;
      w.dx = l == 0 ? dx : ws + S.ws_dx[l];
      w.want_dx = w.dx != nullptr ? 1 : 0;
      w.sxT = g.sxT, w.sxB = g.sxB, w.I = g.I;
      w.syT = g.syT, w.syB = g.syB, w.H = g.H, w.Hg = g.Hg;
    }
    {
      Scope sc(3, s);
      if ((rc = hip_fail(launch_wf_bwd(S.g[0], a, s), "wf_bwd")) != 0) return rc;
    }
  }
  if (wave) {   // the batched half of every layer's backward: one launch each for the whole stack
    WghArgs wh[WF_MAXL];
    RefP rps[WF_MAXL];
    RefG ogs[WF_MAXL];
    const float* wparts[WF_MAXL];
    float* cgs[WF_MAXL];
    const float* ccgs[WF_MAXL];
    for (int l = 0; l < L; ++l) {
      const Layout& Lr = S.lay[l];
      const float* rs = (const float*)ly[l].reserve;
      float* wl = ws + S.ws_layer[l];
      WghArgs& w = wh[l];
      w.dpre = wl + Lr.b_dpre, w.x = l == 0 ? x : (ly[l - 1].drop != nullptr ? ly[l - 1].drop->y_dropped : ly[l - 1].y);
      w.y = ly[l].y, w.h0 = ly[l].h0, w.qx = rs + Lr.r_qx, w.dqx = wl + Lr.b_dqx;
      w.Qs = rs + Lr.r_Qs, w.dQs = wl + Lr.b_dQs, w.wpart = wl + Lr.b_wpart;
      wparts[l] = wl + Lr.b_wpart, cgs[l] = wl + Lr.b_cgrad, ccgs[l] = wl + Lr.b_cgrad;
      rps[l] = to_refp(ly[l].params);
      const vmlmf_grads* gr = ly[l].grads;
      RefG& og = ogs[l];
      og.dia_x = gr->dia_x, og.dia_h = gr->dia_h, og.u_x = gr->u_x, og.v_x = gr->v_x, og.b_x = gr->b_x;
      og.b_h = gr->b_h, og.u_h0 = gr->u_h[0], og.u_h1 = gr->u_h[1], og.v_h0 = gr->v_h[0], og.v_h1 = gr->v_h[1];
      for (int k = 0; k < 4; ++k) og.wg[k] = gr->w_gate[k], og.ug[k] = gr->u_gate[k], og.bg[k] = gr->b_gate[k];
    }
    {
      Scope sc(5, s);
      if ((rc = hip_fail(launch_wgrad_h_stack(L, S.g, wh, s), "wgrad")) != 0) return rc;
    }
    {   // one finishing launch: a workgroup per hidden unit sums that unit's partial sums once and finishes its gradient entries
      bool fu = g_finish_units;
      for (int l = 0; l < L; ++l) fu = fu && finish_units_ok(S.g[l]);
      if (fu) {
        Scope sc(7, s);
        return hip_fail(launch_finish_units_stack(L, S.g, rps, ogs, hb_top, s, health_word(s), wparts, nullptr), "finish");
      }
    }
    bool ffb = g_ffb > 0;   // (wavefront stacks: 48 - 64 blocks per layer; on only when asked for - measured: DESIGN.md)
    for (int l = 0; l < L; ++l) ffb = ffb && finish_from_blocks_ok(S.g[l]);
    if (ffb) {
      Scope sc(7, s);
      return hip_fail(launch_finish_stack(L, S.g, rps, ccgs, ogs, hb_top, s, health_word(s), wparts, nullptr), "finish");
    }
    {
      Scope sc(6, s);
      if ((rc = hip_fail(launch_reduce_stack(L, S.g, wparts, cgs, s), "reduce")) != 0) return rc;
    }
    {
      Scope sc(7, s);
      if ((rc = hip_fail(launch_finish_stack(L, S.g, rps, ccgs, ogs, hb_top, s, health_word(s)), "finish")) != 0) return rc;
    }
    return 0;
  }
  for (int l = L - 1; l >= 0; --l) {
    const VGeo& g = S.g[l];
    const Layout& Lr = S.lay[l];
    const VPack& P = S.P[l];
    const float* rs = (const float*)ly[l].reserve;
    const float* pack = rs + Lr.r_pack;
    float* wl = ws + S.ws_layer[l];
    const float* xl = l == 0 ? x : ly[l - 1].y;
    if (!wave) {   // the per-layer kernels, chained through the dx buffers
      BwdArgs b;
      b.gates = rs + Lr.r_gates, b.cs = rs + Lr.r_cs, b.c0 = ly[l].c0, b.dy = l == L - 1 ? dy : ws + S.ws_dx[l + 1];
      b.dhT = ly[l].dhT, b.dcT = ly[l].dcT;
      b.VR = pack + P.VR, b.UE = pack + P.UE, b.EH = pack + P.EH, b.VE = pack + P.VE;
      b.dpre = wl + Lr.b_dpre, b.dQs = wl + Lr.b_dQs, b.dh0 = ly[l].dh0, b.dc0 = ly[l].dc0, b.trash = wl + Lr.b_trash;
      b.hd = hb;
      memset(&b.wr, 0, sizeof(b.wr));   // (the tape of a stack launch: its progress words are not this path's)
      {
        Scope sc(3, s);
        if ((g_rec3 & 2) && rec3_bwd_supported(g)) {
          if ((rc = hip_fail(launch_rec3_bwd(g, b, s), "rec3_bwd")) != 0) return rc;
        } else if ((rc = hip_fail(launch_rec_bwd(g, b, s), "rec_bwd")) != 0) return rc;
      }
      WgxArgs wx;
      wx.dpre = wl + Lr.b_dpre, wx.VRX = pack + P.VRX, wx.UXO = pack + P.UXO, wx.EXI = pack + P.EXI;
      wx.dx = l == 0 ? dx : ws + S.ws_dx[l], wx.dqx = wl + Lr.b_dqx;
      if (!(g.foldx && wx.dx == nullptr)) {
        Scope sc(4, s);
        if ((rc = hip_fail(launch_wgrad_x(g, wx, s), "dqx_dx")) != 0) return rc;
      }
    }
    if ((rc = backward_tail(g, Lr, ly[l].params, ly[l].grads, xl, ly[l].y, ly[l].h0, rs, wl, hb, s)) != 0) return rc;
  }
  return 0;
}

int vmlmf_tune_generation(void) { return g_tune_generation; }

int vmlmf_check_status(void) { return take_status(); }

int vmlmf_tune(const char* key, int value) {
  if (key == nullptr) return fail(VMLMF_E_BADARG, "tune: null key");
  const std::string k(key);
  if (k == "rbx") g_rbx = value;
  else if (k == "ffb") g_ffb = value;
  else if (k == "rb") g_rb_mode = value;
  else if (k == "rec3") g_rec3 = value;
  else if (k == "test_wride_spin") g_wride_spin = value < 1 ? WRIDE_SPIN_DEFAULT : value;
  else if (k == "inrow") g_inrow = value;
  else if (k == "adam_guard") g_adam_guard = value;
  else if (k == "clear_health") {   // forget a non-finite gradient no guarded optimizer step has consumed (synchronises the device)
    unsigned* hw = vmlmf_health_word_if_any();
    if (hw != nullptr && hipMemset(hw, 0, sizeof(unsigned)) != hipSuccess) (void)hipGetLastError();
  }
  else if (k == "wring") g_wring = value;
  else if (k == "direct") g_direct = value;
  else if (k == "finish2") g_finish2 = value;
  else if (k == "wride") g_wride_tripped.store(value != 0 ? 0 : 1);   // 0: stand-alone weight-gradient kernel; 1: ride again (where VMLMF_WRIDE allows)
  else if (k == "rb_min_batch") g_rb_minB = value < 0 ? 0 : value;   // (0 = never, the default)
  else if (k == "rb_cluster") g_rb_S = value < 0 ? 0 : value;
  else if (k == "rb_rows") g_rb_rows = value < 0 ? 0 : value;
  else return fail(VMLMF_E_BADARG, "tune: unknown key " + k);
  ++g_tune_generation;
  return 0;
}

int vmlmf_tune_get(const char* key, int* value) {
  if (key == nullptr || value == nullptr) return fail(VMLMF_E_BADARG, "tune_get: null pointer");
  const std::string k(key);
  if (k == "rbx") *value = g_rbx;
  else if (k == "ffb") *value = g_ffb;
  else if (k == "rb") *value = g_rb_mode;
  else if (k == "rec3") *value = g_rec3;
  else if (k == "inrow") *value = g_inrow;
  else if (k == "adam_guard") *value = g_adam_guard;
  else if (k == "wring") *value = g_wring;
  else if (k == "direct") *value = g_direct;
  else if (k == "finish2") *value = g_finish2;
  else if (k == "wride") *value = (g_wride && g_wride_tripped.load() == 0) ? 1 : 0;   // 0 also after a bounded wait gave up (VMLMF_ST_WRIDE)
  else if (k == "rb_min_batch") *value = g_rb_minB;
  else if (k == "rb_cluster") *value = g_rb_S;
  else if (k == "rb_rows") *value = g_rb_rows;
  else return fail(VMLMF_E_BADARG, "tune_get: unknown key " + k);
  return 0;
}

int vmlmf_profile_enable(int mask) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.mask = (unsigned)mask;
  return 0;
}

int vmlmf_head_forward(int B, int H, int C, const float* h, long long ldh, const float* weight,
                       const float* bias, float* logits, void* stream) {
  if (B < 1 || H < 1 || C < 1 || ldh < H) return fail(VMLMF_E_BADARG, "head: B, H, C must be >= 1 and ldh >= H");
  if (C > head_max_classes()) return fail(VMLMF_E_UNSUPPORTED, "head: more than 32 classes");
  if (h == nullptr || weight == nullptr || logits == nullptr) return fail(VMLMF_E_BADARG, "head: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Scope sc(8, s);
  hipError_t e = launch_head_fwd(B, H, C, h, ldh, weight, bias, logits, s);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int vmlmf_head_backward(int B, int H, int C, const float* h, long long ldh, const float* weight,
                        const float* dlogits, float* dh, float* dweight, float* dbias, void* stream) {
  if (B < 1 || H < 1 || C < 1 || ldh < H) return fail(VMLMF_E_BADARG, "head: B, H, C must be >= 1 and ldh >= H");
  if (C > head_max_classes()) return fail(VMLMF_E_UNSUPPORTED, "head: more than 32 classes");
  if (h == nullptr || weight == nullptr || dlogits == nullptr) return fail(VMLMF_E_BADARG, "head: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Scope sc(9, s);
  hipError_t e = launch_head_bwd(B, H, C, h, ldh, weight, dlogits, dh, dweight, dbias, s);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int vmlmf_ce_forward(int B, int C, const float* logits, const int64_t* target, int64_t ignore_index, float* loss,
                     float* lse, float* nvalid, float* dlogits_unit, void* stream) {
  if (B < 1 || C < 1) return fail(VMLMF_E_BADARG, "ce: B and C must be >= 1");
  if (!logits || !target || !loss || !lse || !nvalid) return fail(VMLMF_E_BADARG, "ce: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Scope sc(10, s);
  hipError_t e = launch_ce_fwd(B, C, logits, (const long long*)target, (long long)ignore_index, loss, lse, nvalid,
                               dlogits_unit, s);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int vmlmf_ce_backward(int B, int C, const float* logits, const int64_t* target, int64_t ignore_index,
                      const float* lse, const float* nvalid, const float* dloss, float* dlogits, void* stream) {
  if (B < 1 || C < 1) return fail(VMLMF_E_BADARG, "ce: B and C must be >= 1");
  if (!logits || !target || !lse || !nvalid || !dloss || !dlogits) return fail(VMLMF_E_BADARG, "ce: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Scope sc(11, s);
  hipError_t e = launch_ce_bwd(B, C, logits, (const long long*)target, (long long)ignore_index, lse, nvalid, dloss,
                               dlogits, s);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int vmlmf_nll_forward(int R, int V, const float* scores, const int64_t* y, float scale, float* loss, float* lse,
                      float* rowloss, void* stream) {
  if (R < 1 || V < 1) return fail(VMLMF_E_BADARG, "nll: R and V must be >= 1");
  if (!scores || !y || !loss || !lse || !rowloss) return fail(VMLMF_E_BADARG, "nll: null pointer");
  hipError_t e = launch_nll_fwd(R, V, scores, (const long long*)y, scale, loss, lse, rowloss, (hipStream_t)stream);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int vmlmf_nll_backward(int R, int V, const float* scores, const int64_t* y, float scale, const float* lse,
                       const float* dloss, float* dscores, void* stream) {
  if (R < 1 || V < 1) return fail(VMLMF_E_BADARG, "nll: R and V must be >= 1");
  if (!scores || !y || !lse || !dloss || !dscores) return fail(VMLMF_E_BADARG, "nll: null pointer");
  hipError_t e = launch_nll_bwd(R, V, scores, (const long long*)y, scale, lse, dloss, dscores, (hipStream_t)stream);
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

size_t vmlmf_nll_grad_scratch_floats(int R, int V) { return (size_t)nll_grad_workgroups(R < 1 ? 1 : R) * (size_t)(V < 1 ? 1 : V); }

int vmlmf_nll_forward_grad(int R, int V, float* scores, const float* bias, const int64_t* y, float scale, float* loss,
                           float* rowloss, float* dbias, float* scratch, void* stream) {
  if (R < 1 || V < 1) return fail(VMLMF_E_BADARG, "nll: R and V must be >= 1");
  if (!scores || !y || !loss || !rowloss || !scratch) return fail(VMLMF_E_BADARG, "nll: null pointer");
  const int rc = launch_nll_fwd_grad(R, V, scores, bias, (const long long*)y, scale, loss, rowloss, dbias, scratch, (hipStream_t)stream);
  if (rc == -3) return fail(VMLMF_E_UNSUPPORTED, "nll_forward_grad: rows must be 16-byte aligned, a multiple of four and at most 12288 wide");
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

size_t vmlmf_embed_backward_scratch_bytes(int R, int V) { return embed_bwd_scratch_bytes(R < 1 ? 1 : R, V < 1 ? 1 : V); }

int vmlmf_embed_backward(int R, int H, int V, const int64_t* tokens, const float* dy, float* dweight, void* scratch,
                         size_t scratch_bytes, void* stream) {
  if (R < 1 || H < 1 || V < 1) return fail(VMLMF_E_BADARG, "embed: R, H, V must be >= 1");
  if (!tokens || !dy || !dweight) return fail(VMLMF_E_BADARG, "embed: null pointer");
  const int rc = launch_embed_bwd(R, H, V, (const long long*)tokens, dy, dweight, scratch, scratch_bytes, (hipStream_t)stream);
  if (rc == -3) return fail(VMLMF_E_UNSUPPORTED, "embed_backward: embedding width > 1024");
  if (rc == -4) return fail(VMLMF_E_WORKSPACE, "embed_backward: scratch smaller than vmlmf_embed_backward_scratch_bytes()");
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

// ---- dropout of the LM network (ABI 11; vmlmf_dropout.h) ----
int vmlmf_dropout_fused(const vmlmf_desc* d) {
  VGeo g;
  RbGeo q;
  if (make_geo(d, &g, &q) != 0) return 0;
  return (g.rb && g.syT == (long long)g.B * g.H) ? 1 : 0;
}

int vmlmf_dropout_advance(int64_t* state, int64_t* snapshot, void* stream) {
  if (!state || !snapshot || state == snapshot) return fail(VMLMF_E_BADARG, "dropout_advance: two distinct device words pairs");
  const int rc = launch_drop_advance(reinterpret_cast<unsigned long long*>(state), reinterpret_cast<unsigned long long*>(snapshot), (hipStream_t)stream);
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

static int drop_rows(int mode, int64_t R, int H, int V, float p, const int64_t* state, int site, const DropCols& cm, const float* x,
                     const int64_t* tokens, float* y, void* stream) {
  if (R < 0 || H < 1) return fail(VMLMF_E_BADARG, "dropout: R >= 0, H >= 1");
  if (!(p >= 0.f && p < 1.f)) return fail(VMLMF_E_BADARG, "dropout: p must be in [0, 1)");
  if (!state || !y || (mode != 1 && !x) || (mode == 2 && !tokens)) return fail(VMLMF_E_BADARG, "dropout: null pointer");
  if (R >= (1ll << 32)) return fail(VMLMF_E_UNSUPPORTED, "dropout: 2^32 positions and more");
  DropArgs d;
  memset(&d, 0, sizeof(d));
  d.state = reinterpret_cast<const unsigned long long*>(state), d.thresh = drop_thresh(p), d.scale = 1.f / (1.f - p), d.site = site;
  const int rc = launch_drop_rows(mode, R, H, V, d, cm, x, (const long long*)tokens, y, (hipStream_t)stream);
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

int vmlmf_dropout_apply(int64_t R, int H, const float* x, float* y, float p, const int64_t* state, int site, void* stream) {
  const DropCols cm = {H, 0};
  return drop_rows(0, R, H, 0, p, state, site, cm, x, nullptr, y, stream);
}

int vmlmf_dropout_factors(const vmlmf_desc* d, int64_t R, int H, float p, const int64_t* state, int site, float* factors, void* stream) {
  DropCols cm = {H, 0};
  if (d != nullptr) {
    VGeo g;
    RbGeo q;
    int rc = make_geo(d, &g, &q);
    if (rc != 0) return rc;
    if (g.H != H) return fail(VMLMF_E_BADARG, "dropout_factors: H is not the layer's hidden size");
    if (g.rb) cm.Hg = g.Hg, cm.gstride = 64 * g.W;
  }
  return drop_rows(1, R, H, 0, p, state, site, cm, nullptr, nullptr, factors, stream);
}

int vmlmf_embed_dropout_forward(int R, int H, int V, const int64_t* tokens, const float* weight, float* out, float p, const int64_t* state,
                                int site, void* stream) {
  if (V < 1) return fail(VMLMF_E_BADARG, "embed: V must be >= 1");
  const DropCols cm = {H, 0};
  return drop_rows(2, R, H, V, p, state, site, cm, weight, tokens, out, stream);
}

int vmlmf_embed_dropout_backward(int R, int H, int V, const int64_t* tokens, const float* dy, float* dweight, void* scratch,
                                 size_t scratch_bytes, float p, const int64_t* state, int site, void* stream) {
  if (R < 1 || H < 1 || V < 1) return fail(VMLMF_E_BADARG, "embed: R, H, V must be >= 1");
  if (!tokens || !dy || !dweight || !state) return fail(VMLMF_E_BADARG, "embed: null pointer");
  if (!(p >= 0.f && p < 1.f)) return fail(VMLMF_E_BADARG, "dropout: p must be in [0, 1)");
  DropArgs d;
  memset(&d, 0, sizeof(d));
  d.state = reinterpret_cast<const unsigned long long*>(state), d.thresh = drop_thresh(p), d.scale = 1.f / (1.f - p), d.site = site;
  const int rc = launch_embed_bwd(R, H, V, (const long long*)tokens, dy, dweight, scratch, scratch_bytes, (hipStream_t)stream, &d);
  if (rc == -3) return fail(VMLMF_E_UNSUPPORTED, "embed_dropout_backward: embedding width > 1024");
  if (rc == -4) return fail(VMLMF_E_WORKSPACE, "embed_backward: scratch smaller than vmlmf_embed_backward_scratch_bytes()");
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

int vmlmf_transpose(int rows, int cols, const float* src, float* dst, void* stream) {
  if (rows < 1 || cols < 1) return fail(VMLMF_E_BADARG, "transpose: rows, cols must be >= 1");
  if (!src || !dst || src == dst) return fail(VMLMF_E_BADARG, "transpose: two distinct buffers");
  const int rc = launch_transpose(rows, cols, src, dst, (hipStream_t)stream);
  return rc == 0 ? 0 : fail(rc, hipGetErrorString((hipError_t)rc));
}

int vmlmf_profile_read(float* usec_sum, int32_t* count, int reset) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (int k = 0; k < NKERN; ++k) {
    float sum = 0.f;
    for (auto& pr : g_prof.ev[k]) {
      (void)hipEventSynchronize(pr.second);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, pr.first, pr.second);
      sum += ms * 1000.f;
    }
    if (usec_sum != nullptr) usec_sum[k] = sum;
    if (count != nullptr) count[k] = (int32_t)g_prof.ev[k].size();
    if (reset) {
      for (auto& pr : g_prof.ev[k]) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
      g_prof.ev[k].clear();
    }
  }
  return 0;
}

const char* vmlmf_kernel_name(int k) {
  return kernel_label(k);
}

}  // extern "C"

namespace {
const char* kernel_label(int k) {
  static const char* names[NKERN] = {"pack_kernel",    "xproj_kernel",   "rec_fwd_kernel", "rec_bwd_kernel",
                                     "dqx_dx_kernel", "wgrad_mfma_kernel", "reduce_cg_kernel",  "finish_kernel",
                                     "head_fwd_kernel", "head_bwd_kernel", "ce_fwd_kernel", "ce_bwd_kernel", "finish2_kernel"};
  return (k >= 0 && k < NKERN) ? names[k] : "";
}
}  // namespace
