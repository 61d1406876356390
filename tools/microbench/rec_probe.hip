// Phase-timing probe for rec_fwd_kernel at the bench shape (B=64 T=128 H=180 rank 16): s_memtime stamps at
// the phase boundaries of the compute waves.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vmlmf_amd/csrc
//   tools/microbench/rec_probe.hip -o tools/microbench/bin/rec_probe ; run on the GPU box.  Not part of the library.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vmlmf_rec_fwd.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int ABL>
float run(const VGeo& g, const FwdArgs& a, int iters) {
  constexpr int KQ = 16;
  const size_t lds = sizeof(float) * ((size_t)2 * g.NW * KQ + (size_t)FWD_NB * g.NT * 4 + (size_t)2 * g.NT * 8 + 256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((rec_fwd_kernel<16, 1, false, 256, 3, ABL>), dim3(g.nwg), dim3(g.NT + 128), lds, 0, g, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((rec_fwd_kernel<16, 1, false, 256, 3, ABL>), dim3(g.nwg), dim3(g.NT + 128), lds, 0, g, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main() {
  VGeo g = {};
  g.variant = 1, g.B = 64, g.T = 128, g.I = 9, g.H = 180, g.rw = 16, g.G = 1, g.Hg = 180, g.W = 3, g.NT = 192,
  g.NW = 3, g.ru0 = 16, g.off1 = 16, g.KX = 16, g.KH = 16, g.NP = 1, g.KQ = 16, g.NPX = 1, g.KQX = 16, g.R = 1,
  g.nwg = 64, g.Bp = 64, g.syT = 180, g.syB = 128 * 180, g.sxT = 9, g.sxB = 128 * 9;
  const size_t TS = (size_t)g.T * g.Bp * g.NT;
  auto dalloc = [&](size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
  };
  FwdArgs a = {};
  a.gx = dalloc(TS * 4, 1.0f);
  a.VE = dalloc((size_t)4 * 16 * g.NT, 0.2f);
  a.UR = dalloc((size_t)16 * g.NT, 0.2f);
  a.EH = dalloc((size_t)4 * g.NT, 0.2f);
  a.h0 = nullptr, a.c0 = nullptr;
  a.y = dalloc((size_t)g.B * g.T * g.H, 0.f);
  a.hT = dalloc((size_t)g.B * g.H, 0.f);
  a.cT = dalloc((size_t)g.B * g.H, 0.f);
  a.gates = dalloc(TS * 4, 0.f);
  a.cs = dalloc(TS + (size_t)g.Bp * g.NT, 0.f);
  a.Qs = dalloc((size_t)g.T * g.B * 16, 0.f);
  a.trash = dalloc(256, 0.f);
  printf("full                         %8.2f us\n", run<0>(g, a, 50));
  printf("instrumented          (256)  %8.2f us\n", run<256>(g, a, 5));
  float hbuf[256];
  CK(hipMemcpy(hbuf, a.trash, sizeof(hbuf), hipMemcpyDeviceToHost));
  const char* names[6] = {"reduce+write", "wait+barrier", "lds sum", "fma", "gates+outs", "step total"};
  for (int w = 0; w < 3; ++w) {
    printf("wave %d memtime ticks/step:", w);
    for (int i = 0; i < 6; ++i) printf("  %s %.0f", names[i], hbuf[64 + w * 8 + i]);
    printf("\n");
  }
  return 0;
}
