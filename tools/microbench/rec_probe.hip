// Phase-timing probe for rec_fwd_kernel at the bench shape (B=64 T=128 H=180 rank 16): s_memtime stamps at
// the phase boundaries of the compute waves.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vmlmf_amd/csrc
//   tools/microbench/rec_probe.hip -o tools/microbench/bin/rec_probe ; run on the GPU box.  Not part of the library.
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef VMLMF_STW7
#define MOVER_THREADS 320
#else
#define MOVER_THREADS 128
#endif
#include "vmlmf_rec3.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int ABL, bool XW = false>
float run(const VGeo& g, const FwdArgs& a, const XwArgs& xw, int iters) {
  constexpr int KQ = 16;
  const size_t lds = sizeof(float) * ((size_t)2 * g.NW * KQ + (size_t)FWD_NB * g.NT * 4 + (size_t)2 * g.NT * 8 + 256 + 4 +
                                      xwave_lds_floats());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((rec_fwd_kernel<16, 1, false, 256, 3, ABL, XW>), dim3(g.nwg), dim3(g.NT + MOVER_THREADS), lds, 0, g, a, xw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((rec_fwd_kernel<16, 1, false, 256, 3, ABL, XW>), dim3(g.nwg), dim3(g.NT + MOVER_THREADS), lds, 0, g, a, xw);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <int ABL, int XM>
float run3(const VGeo& g, const FwdArgs& a, const XwArgs& xw, int iters) {
  const size_t lds = sizeof(float) * rec3_fwd_lds_floats_xm(g.NT, g.T, 12, XM);
  auto kern = rec3_fwd_kernel<16, 12, ABL, XM>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(g.nwg), dim3(g.NT + 64), lds, 0, g, a, xw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(g.nwg), dim3(g.NT + 64), lds, 0, g, a, xw);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <int ABL>
float run_bwd(const VGeo& g, const BwdArgs& b, int iters) {
  const size_t lds = sizeof(float) * ((size_t)2 * g.NW * 16 + (size_t)BWD_NB * g.NT * 6 + (size_t)2 * g.NT * 4 + 256 + 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((rec_bwd_kernel<16, 1, false, 256, 3, ABL>), dim3(g.nwg), dim3(g.NT + MOVER_THREADS), lds, 0, g, b);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((rec_bwd_kernel<16, 1, false, 256, 3, ABL>), dim3(g.nwg), dim3(g.NT + MOVER_THREADS), lds, 0, g, b);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <int ABL>
float run3_bwd(const VGeo& g, const BwdArgs& b, int iters) {
  const size_t lds = sizeof(float) * rec3_bwd_lds_floats(g.NT);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rec3_bwd_kernel<16, ABL>), dim3(g.nwg), dim3(g.NT + 128), lds, 0, g, b);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rec3_bwd_kernel<16, ABL>), dim3(g.nwg), dim3(g.NT + 128), lds, 0, g, b);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
  VGeo g = {};
  g.variant = 1, g.B = 64, g.T = 128, g.I = 9, g.H = 180, g.rw = 16, g.G = 1, g.Hg = 180, g.W = 3, g.NT = 192,
  g.NW = 3, g.ru0 = 16, g.off1 = 16, g.KX = 16, g.KH = 16, g.NP = 1, g.KQ = 16, g.NPX = 1, g.KQX = 16, g.R = 1,
  g.nwg = 64, g.Bp = 64, g.syT = 180, g.syB = 128 * 180, g.sxT = 9, g.sxB = 128 * 9;
  if (argc > 2) { g.B = g.Bp = g.nwg = atoi(argv[2]); g.syB = 128 * 180; }   // fewer rows: smaller per-step strides
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
  a.gates = dalloc((TS + (size_t)g.Bp * g.NT) * 4, 0.f);
  a.cs = dalloc(TS + (size_t)g.Bp * g.NT, 0.f);
  a.Qs = dalloc((size_t)g.T * g.B * 16, 0.f);
  a.trash = dalloc(512, 0.f);
  XwArgs xw = {};
  xw.x = dalloc((size_t)g.B * g.T * g.I, 1.0f);
  xw.UXP = dalloc((size_t)g.I * g.KX, 0.2f);
  xw.WXD = dalloc((size_t)4 * g.I * g.NT, 0.2f);
  xw.BBT = dalloc((size_t)4 * g.H, 0.2f);
  a.qxw = dalloc((size_t)g.T * g.B * g.KX, 0.f);
  const bool xwave = argc > 1;
  a.xwave = xwave ? 1 : 0;
  if (argc > 1 && argv[1][0] == '3') {
    printf("rec3 fwd XM0 full            %8.2f us\n", (run3<0, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM0 no x            %8.2f us\n", (run3<1, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM0 storer idle     %8.2f us\n", (run3<2048, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM0 both            %8.2f us\n", (run3<2049, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM0 narrow c/y st   %8.2f us\n", (run3<64, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM0 narrow, no x    %8.2f us\n", (run3<65, 0>(g, a, xw, 50)));
    printf("rec3 fwd XM1 full            %8.2f us\n", (run3<0, 1>(g, a, xw, 50)));
    printf("rec3 fwd XM1 no x            %8.2f us\n", (run3<1, 1>(g, a, xw, 50)));
    printf("rec3 fwd XM1 storer idle     %8.2f us\n", (run3<2048, 1>(g, a, xw, 50)));
    printf("rec3 fwd XM2 full            %8.2f us\n", (run3<0, 2>(g, a, xw, 50)));
    printf("rec3 fwd XM2 no x            %8.2f us\n", (run3<1, 2>(g, a, xw, 50)));
    printf("rec3 fwd XM2 storer idle     %8.2f us\n", (run3<2048, 2>(g, a, xw, 50)));
    {
      BwdArgs b = {};
      b.gates = a.gates, b.cs = a.cs, b.dy = nullptr, b.dhT = a.hT, b.dcT = nullptr;
      b.VR = dalloc((size_t)4 * 16 * g.NT, 0.2f), b.UE = dalloc((size_t)16 * g.NT, 0.2f), b.EH = a.EH, b.VE = a.VE;
      b.dpre = dalloc(TS * 4, 0.f), b.dQs = dalloc((size_t)g.T * g.B * 16, 0.f), b.trash = a.trash;
      printf("rec_bwd  full                %8.2f us\n", run_bwd<0>(g, b, 50));
      printf("rec3 bwd full                %8.2f us\n", run3_bwd<0>(g, b, 50));
      printf("rec3 bwd, storer idle        %8.2f us\n", run3_bwd<2048>(g, b, 50));
      printf("rec3 bwd, loader idle        %8.2f us\n", run3_bwd<4096>(g, b, 50));
      printf("rec3 bwd, both idle          %8.2f us\n", run3_bwd<6144>(g, b, 50));
      printf("rec3 bwd instrumented (256)  %8.2f us\n", run3_bwd<256>(g, b, 20));
      float tb[64];
      CK(hipMemcpy(tb, a.trash + 64, sizeof(tb), hipMemcpyDeviceToHost));
      for (int w = 0; w < 3; ++w)
        printf("rec3 bwd wave %d memtime ticks/step:  derivs+fold %.0f  rotate-add+rowsum+write %.0f  prepare %.0f  barrier %.0f  read+total+expansion %.0f  step total %.0f\n",
               w, tb[w * 8 + 0], tb[w * 8 + 1], tb[w * 8 + 2], tb[w * 8 + 3], tb[w * 8 + 4], tb[w * 8 + 5]);
    }
    return 0;
  }
  if (xwave) {
    printf("x-wave full                  %8.2f us\n", (run<0, true>(g, a, xw, 50)));
    printf("x-wave, storer: no stores    %8.2f us\n", (run<1024, true>(g, a, xw, 50)));
    printf("x-wave, storer idle          %8.2f us\n", (run<2048, true>(g, a, xw, 50)));
    printf("x-wave without its pk FMAs   %8.2f us\n", (run<4096, true>(g, a, xw, 50)));
    printf("x-wave without FMAs and qx   %8.2f us\n", (run<12288, true>(g, a, xw, 50)));
    printf("no stores AND no x FMAs      %8.2f us\n", (run<1024 + 4096, true>(g, a, xw, 50)));
    printf("storer idle AND no x FMAs    %8.2f us\n", (run<2048 + 4096, true>(g, a, xw, 50)));
    printf("x-wave instrumented   (256)  %8.2f us\n", (run<256, true>(g, a, xw, 5)));
  } else {
    printf("full                         %8.2f us\n", (run<0>(g, a, xw, 50)));
    printf("storer: no global stores     %8.2f us\n", (run<1024>(g, a, xw, 50)));
    printf("storer: idle                 %8.2f us\n", (run<2048>(g, a, xw, 50)));
    printf("idle storer, no DPP reduce   %8.2f us\n", (run<2048 + 1>(g, a, xw, 50)));
    printf("idle storer, 1/4 of the FMAs %8.2f us\n", (run<2048 + 2>(g, a, xw, 50)));
    printf("idle storer, no exp/rcp      %8.2f us\n", (run<2048 + 4>(g, a, xw, 50)));
    printf("idle storer, all three       %8.2f us\n", (run<2048 + 7>(g, a, xw, 50)));
    printf("instrumented          (256)  %8.2f us\n", (run<256>(g, a, xw, 5)));
  }
  {
    BwdArgs b = {};
    b.gates = a.gates, b.cs = a.cs, b.dy = nullptr, b.dhT = a.hT, b.dcT = nullptr;
    b.VR = dalloc((size_t)4 * 16 * g.NT, 0.2f), b.UE = dalloc((size_t)16 * g.NT, 0.2f), b.EH = a.EH;
    b.dpre = dalloc(TS * 4, 0.f), b.dQs = dalloc((size_t)g.T * g.B * 16, 0.f), b.trash = a.trash;
    printf("rec_bwd full                 %8.2f us\n", run_bwd<0>(g, b, 50));
    printf("rec_bwd, storer idle         %8.2f us\n", run_bwd<2048>(g, b, 50));
    printf("rec_bwd, loader idle         %8.2f us\n", run_bwd<4096>(g, b, 50));
    printf("rec_bwd, both idle           %8.2f us\n", run_bwd<6144>(g, b, 50));
  }
  if (xwave) {
    run<65536, true>(g, a, xw, 1);
    float tb[64];
    CK(hipMemcpy(tb, a.trash + 128, sizeof(tb), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k)
      printf("barrier %d arrival (ticks, relative to compute wave 0): w1 %+.0f  w2 %+.0f  storer %+.0f\n", 65 + k,
             tb[8 + 2 * k] - tb[2 * k], tb[16 + 2 * k] - tb[2 * k], tb[32 + 2 * k] - tb[2 * k]);
  }
  float hbuf[256];
  CK(hipMemcpy(hbuf, a.trash, sizeof(hbuf), hipMemcpyDeviceToHost));
  const char* names[6] = {"reduce+write", "wait+barrier", "lds sum", "fma", "gates+outs", "step total"};
  for (int w = 0; w < 3; ++w) {
    printf("wave %d memtime ticks/step:", w);
    for (int i = 0; i < 6; ++i) printf("  %s %.0f", names[i], hbuf[64 + w * 8 + i]);
    printf("\n");
  }
  printf("loader: issue %.0f  vmcnt wait %.0f   storer: busy %.0f   (ticks per step)\n", hbuf[64 + 24], hbuf[64 + 25], hbuf[64 + 26]);
  return 0;
}
