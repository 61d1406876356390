// Dropout of the LM network (V/src/models/vmlmf_lm.py:434-439: `x = self.dropout(x)` behind the embedding and behind every LSTM
// layer, nn.Dropout(p) of :402) without mask tensors and, on the layer kernels that take it, without launches of its own (verdict
// r4 item 3): the keep / drop decision of an element is a pure function of (seed, offset, site, position, column) - Philox4x32-10
// (Salmon et al., SC'11; the counter-based generator the stock op uses too), one call per four neighbouring columns - so the
// forward kernel that stores an activation writes its dropped copy beside it, and the backward kernel that reads the upstream
// gradient of that copy regenerates the same bits.
//   counter = (position, column >> 2, site, offset low word), key = (seed low word, seed high word + offset high word)
//   element (position, column) is DROPPED iff word[column & 3] < thresh, thresh = round(p 2^32); kept values are scaled by 1/(1-p)
//   position = t B + b (the row of the flattened (T, B) grid), site = 0 for the embedding's output, l + 1 for layer l's
//   column   = the hidden unit, or - inside the row-block kernels - the unit's thread slot (vmlmf_geo.h: group g's units start at
//              slot 64 W g, a multiple of four, so the four units a lane owns are one call; DropCols below is that map)
// (seed, offset) live in device memory: a captured graph replays with fresh masks because vmlmf_dropout_advance - a node of the
// same graph - snapshots the pair and increments the offset.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DropArgs {
  const unsigned long long* state;   // {seed, offset} snapshot of this forward; nullptr: no dropout
  float* yd;                         // forward of a layer: the dropped copy of y (same layout)
  unsigned thresh;                   // dropped iff word < thresh
  float scale;                       // 1 / (1 - p)
  int site, pad;
};
// hidden unit -> column of the counter: (n / Hg) * gstride + n % Hg; identity: Hg = H, gstride = 0
struct DropCols {
  int Hg, gstride;
};

struct DropKey {
  unsigned k0, k1, c2, c3;
};
__device__ __forceinline__ DropKey drop_key(const DropArgs& d) {
  const unsigned long long seed = d.state[0], off = d.state[1];
  DropKey k;
  k.k0 = (unsigned)seed, k.k1 = (unsigned)(seed >> 32) + (unsigned)(off >> 32), k.c2 = (unsigned)d.site, k.c3 = (unsigned)off;
  return k;
}
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1, c3 = (unsigned)p0, c0 = n0, c2 = n2;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
// the four factors (0 or scale) of columns 4 quad .. 4 quad + 3 at `position`
__device__ __forceinline__ void drop_factors(const DropKey& k, const unsigned thresh, const float scale, const unsigned position, const unsigned quad,
                                             float (&f)[4]) {
  unsigned w[4];
  philox4x32_10(position, quad, k.c2, k.c3, k.k0, k.k1, w);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = w[i] < thresh ? 0.f : scale;
}

// host side of DropArgs: p in [0, 1)
inline unsigned drop_thresh(float p) {
  const double t = (double)p * 4294967296.0 + 0.5;
  return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}
