// Device-side helpers: reference-layout accessors, DPP row rotation, wave reductions, fast gates.
// gfx950 (CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "vmlmf_geo.h"

// Parameter pointers in the reference's layouts (include/vmlmf_hip.h: vmlmf_params).
struct RefP {
  const float *dia_x, *dia_h, *u_x, *v_x, *b_x, *b_h, *u_h0, *u_h1, *v_h0, *v_h1;
  const float *wg[4], *ug[4], *bg[4];   // V5: per-gate tensors
};
struct RefG {
  float *dia_x, *dia_h, *u_x, *v_x, *b_x, *b_h, *u_h0, *u_h1, *v_h0, *v_h1;
  float *wg[4], *ug[4], *bg[4];
};

// ---------------------------------------------------------------------------------------------------
// compile-time loops (DPP controls must be immediates)
// ---------------------------------------------------------------------------------------------------
template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// Rotate a value by K lanes inside each 16-lane DPP row (row_ror:K).  Which neighbour a lane receives
// from is calibrated at pack time by applying the same instruction to lane ids (pack_kernel), so the
// register images match the hardware's direction by construction.
template <int K>
__device__ __forceinline__ float ror16(float v) {
  if constexpr (K == 0) {
    return v;
  } else {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + K, 0xf, 0xf, true));
  }
}

// acc += (value received through row_ror:K) * w   as ONE instruction (v_fmac_f32 with a DPP source operand).
// hipcc does not fold update_dpp into the FMA (it emits v_mov_dpp + s_nop + v_fmac), so this is inline asm.
// DPP hazard (VALU write of `src` -> DPP read needs 2 wait states): callers pass `src` through dpp_fence()
// once before a block of these, which also orders the block after the producer of `src`.
template <int K>
__device__ __forceinline__ void fmac_ror(float& acc, float src, float w) {
  if constexpr (K == 0) {
    acc = fmaf(src, w, acc);
  } else {
    asm("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(K));
  }
}
__device__ __forceinline__ void dpp_fence(float& src) { asm volatile("s_nop 1" : "+v"(src)); }

// Whole rotation chains as ONE asm statement.  Between two single-instruction asm statements that write and then use the
// same accumulator hipcc inserts an s_nop (it cannot know that the accumulator is not the DPP operand): one per pair of
// instructions in the forward reduce, one per group of four in the backward one - 8 to 16 issue slots of a step.
#define VG_DPP1(acc, src, w, k) "v_fmac_f32_dpp %" #acc ", %" #src ", %" #w " row_ror:" #k " row_mask:0xf bank_mask:0xf\n\t"
// 16 rotations of `src` against w[0..15], even rotations into a0, odd ones into a1
__device__ __forceinline__ void fmac_ror_x16(float& a0, float& a1, float src, const float* w) {
  asm("v_fmac_f32 %0, %2, %3\n\t" VG_DPP1(1, 2, 4, 1) VG_DPP1(0, 2, 5, 2) VG_DPP1(1, 2, 6, 3) VG_DPP1(0, 2, 7, 4) VG_DPP1(1, 2, 8, 5)
      VG_DPP1(0, 2, 9, 6) VG_DPP1(1, 2, 10, 7) VG_DPP1(0, 2, 11, 8) VG_DPP1(1, 2, 12, 9) VG_DPP1(0, 2, 13, 10) VG_DPP1(1, 2, 14, 11)
      VG_DPP1(0, 2, 15, 12) VG_DPP1(1, 2, 16, 13) VG_DPP1(0, 2, 17, 14) VG_DPP1(1, 2, 18, 15)
      : "+v"(a0), "+v"(a1)
      : "v"(src), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
        "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
}
// 8 rotations (0..7) against w[0..7] (the half pass of vmlmf_wave.inc)
__device__ __forceinline__ void fmac_ror_x8(float& a0, float& a1, float src, const float* w) {
  asm("v_fmac_f32 %0, %2, %3\n\t" VG_DPP1(1, 2, 4, 1) VG_DPP1(0, 2, 5, 2) VG_DPP1(1, 2, 6, 3) VG_DPP1(0, 2, 7, 4) VG_DPP1(1, 2, 8, 5)
      VG_DPP1(0, 2, 9, 6) VG_DPP1(1, 2, 10, 7)
      : "+v"(a0), "+v"(a1)
      : "v"(src), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
}
// four rotations K0 .. K0+3 of the four gate derivatives d[0..3] against w_g[K0 .. K0+3], gate g into acc[g] (interleaved:
// four independent chains)
#define VG_STEP4(k, w0, w1, w2, w3) VG_DPP1(0, 4, w0, k) VG_DPP1(1, 5, w1, k) VG_DPP1(2, 6, w2, k) VG_DPP1(3, 7, w3, k)
#define VG_STEP4_PLAIN(w0, w1, w2, w3) \
  "v_fmac_f32 %0, %4, %" #w0 "\n\tv_fmac_f32 %1, %5, %" #w1 "\n\tv_fmac_f32 %2, %6, %" #w2 "\n\tv_fmac_f32 %3, %7, %" #w3 "\n\t"
#define VG_BLOCK4_OPERANDS                                                                                                  \
  : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                                   \
  : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(w0[K0]), "v"(w1[K0]), "v"(w2[K0]), "v"(w3[K0]), "v"(w0[K0 + 1]), "v"(w1[K0 + 1]), \
    "v"(w2[K0 + 1]), "v"(w3[K0 + 1]), "v"(w0[K0 + 2]), "v"(w1[K0 + 2]), "v"(w2[K0 + 2]), "v"(w3[K0 + 2]), "v"(w0[K0 + 3]),  \
    "v"(w1[K0 + 3]), "v"(w2[K0 + 3]), "v"(w3[K0 + 3])
template <int K0>
__device__ __forceinline__ void fmac_ror_4x4(float (&acc)[4], const float (&d)[4], const float* w0, const float* w1, const float* w2,
                                             const float* w3) {
  static_assert(K0 == 0 || K0 == 4 || K0 == 8 || K0 == 12, "blocks of four rotations");
  if constexpr (K0 == 0) {
    asm(VG_STEP4_PLAIN(8, 9, 10, 11) VG_STEP4(1, 12, 13, 14, 15) VG_STEP4(2, 16, 17, 18, 19) VG_STEP4(3, 20, 21, 22, 23) VG_BLOCK4_OPERANDS);
  } else if constexpr (K0 == 4) {
    asm(VG_STEP4(4, 8, 9, 10, 11) VG_STEP4(5, 12, 13, 14, 15) VG_STEP4(6, 16, 17, 18, 19) VG_STEP4(7, 20, 21, 22, 23) VG_BLOCK4_OPERANDS);
  } else if constexpr (K0 == 8) {
    asm(VG_STEP4(8, 8, 9, 10, 11) VG_STEP4(9, 12, 13, 14, 15) VG_STEP4(10, 16, 17, 18, 19) VG_STEP4(11, 20, 21, 22, 23) VG_BLOCK4_OPERANDS);
  } else {
    asm(VG_STEP4(12, 8, 9, 10, 11) VG_STEP4(13, 12, 13, 14, 15) VG_STEP4(14, 16, 17, 18, 19) VG_STEP4(15, 20, 21, 22, 23) VG_BLOCK4_OPERANDS);
  }
}

// x + (x rotated by K lanes inside the 16-lane row), one instruction (v_add_f32 with a DPP operand)
template <int K>
__device__ __forceinline__ float add_ror16(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + K, 0xf, 0xf, true));
}
// Sum of the lanes {cc, cc+NC, cc+2NC, ...} of a 16-lane row, delivered to every lane of the class (NC = 2,4,8)
template <int NC>
__device__ __forceinline__ float class_sum16(float x) {
  x = add_ror16<8>(x);
  if constexpr (NC <= 4) x = add_ror16<4>(x);
  if constexpr (NC <= 2) x = add_ror16<2>(x);
  return x;
}
__device__ __forceinline__ float bcast_lane(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

// Sum over the four 16-lane rows of a wave: every lane i ends with v[i] + v[i+16] + v[i+32] + v[i+48].
// sum over each pair of 16-lane rows (rows 0+1 and rows 2+3), every lane of the pair gets it
__device__ __forceinline__ float rowsum2(float v) {
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rowsum4(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// v_exp_f32 / v_rcp_f32 based gates (about 2 ulp; the parity tests bound the end-to-end error).
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2 / (1 + exp(2x)); saturates cleanly for large |x|
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Select between two addresses WITHOUT letting the compiler turn it into two branch-guarded memory ops
// (a memory op under a branch wrecks the counted s_waitcnt vmcnt(N) of the software-pipelined loops).
// The pointer stays in the GLOBAL address space: a generic pointer would become flat_load/flat_store,
// which also tick lgkmcnt and would be drained by the LDS-only barrier wait.
typedef __attribute__((address_space(1))) float gf32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;
__device__ __forceinline__ gf32* sel_g(bool c, float* a, float* b) {
  gf32* p = c ? (gf32*)a : (gf32*)b;
  asm("" : "+v"(p));
  return p;
}
__device__ __forceinline__ const gf32* sel_g(bool c, const float* a, const float* b) {
  const gf32* p = c ? (const gf32*)a : (const gf32*)b;
  asm("" : "+v"(p));
  return p;
}
__device__ __forceinline__ void st4g(gf32* p, float4 v) {
  *reinterpret_cast<gf32x4*>(p) = f32x4{v.x, v.y, v.z, v.w};
}
// (Every inline-asm store of more than 8 bytes is followed by `s_nop 1`: gfx940+ needs two wait states between such a store
// and a VALU write of its data registers; the compiler inserts them for stores it emits itself, but it cannot see into an asm
// block, and did schedule an address computation into a data register right behind one: a wrong first element in a few rows.)
// Store written through to agent scope (sc1): visible to workgroups on every XCD once vmcnt has counted it,
// without the whole-L2 write-back of a release fence (split-K partials of the step-wise GEMMs).  Inline asm: the
// compiler's waitcnt pass does not see it, the caller waits on vmcnt itself.
__device__ __forceinline__ void st4g_agent(gf32* p, float4 v) {
  const f32x4 t = f32x4{v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st1g_agent(gf32* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// Stores in the scalar-base + 32-bit vector byte-offset form (global_store ... v_off, v_data, s[base:base+1]).
// hipcc turns `uniform_ptr[per_lane_index]` into 64-bit per-lane pointers and updates them with vector adds; in
// the storer waves of the recurrent kernels that arithmetic was most of their ~200 vector instructions per step.
// Inline asm: invisible to the compiler's waitcnt pass, which is fine for waves that never wait on stores - and invisible to
// its hazard recognizer, which matters twice on gfx940+:
//  (1) a store of more than 8 bytes needs two wait states before a VALU instruction overwrites its data registers.  The
//      compiler scheduled an address computation into the first data register right behind such a store (a wrong first
//      element of dpre[0] in ~20 % of the rows of a new kernel, timing-dependent): every wide store carries `s_nop 1` behind it.
//  (2) a scalar register written by a VALU instruction (v_readlane: a spilled scalar register coming back from its vector
//      lane) needs five wait states before a vector memory instruction uses it as an address.  In the large kernels the
//      reload sat right in front of the store, which then went to a garbage address (memory access faults that came and went
//      with unrelated changes of a kernel; the faults of the wavefront kernels with a noinline function were the same thing).
//      SAFE = true puts `s_nop 4` in front of the store.  It is not the default: the storer waves are on the critical path of
//      the recurrent kernels and 20 clocks a store cost 14 us of the headline step (a copy of the base through s_mov_b64
//      cost 20, the same stores as raw-buffer builtins - which the compiler does guard by itself - 16: four scalar registers
//      a base, more spills).  The instantiations whose scalar registers do spill around the stores set SAFE (see the
//      SAFE_ST constants of the kernels), and tools/check_asm_hazards.py, run by the CPU tests on the built library,
//      disassembles every kernel and fails on either hazard at any store with a scalar base.
// NT: the non-temporal hint (streamed data nobody re-reads soon: the forward's tape, 41 MB per launch at the headline shape - ten
// times the L2 of an XCD; rec_fwd_kernel's storer 69.1 -> 67.5 us, 0.1566 -> 0.1559 ms per step same-box, tools/sessions/r03y6.sh)
template <bool SAFE = false, bool NT = false>
__device__ __forceinline__ void st4_sv(const void* sbase, unsigned voff, float4 v) {
  const f32x4 t = f32x4{v.x, v.y, v.z, v.w};
  static_assert(!(SAFE && NT), "no instantiation needs both");
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
  else if constexpr (SAFE) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
}
template <bool SAFE = false, bool NT = false>
__device__ __forceinline__ void st1_sv(const void* sbase, unsigned voff, float v) {
  if constexpr (NT) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else if constexpr (SAFE) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// (8 bytes: no wait states needed behind it; non-temporal: tapes)
__device__ __forceinline__ void st2_sv_nt(const void* sbase, unsigned voff, float2 v) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t t = f32x2_t{v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
}
// system-scope write-through variants (sc0 sc1): rows another workgroup, possibly on another XCD, consumes during the launch
template <bool SAFE = false>
__device__ __forceinline__ void st4_sv_sys(const void* sbase, unsigned voff, float4 v) {
  const f32x4 t = f32x4{v.x, v.y, v.z, v.w};
  if constexpr (SAFE) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(t), "s"(sbase) : "memory");
}
template <bool SAFE = false, class V>
__device__ __forceinline__ void st1_sv_sys(const void* sbase, unsigned voff, V v) {   // V: float or unsigned
  if constexpr (SAFE) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else asm volatile("global_store_dword %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// LDS-DMA: 64 lanes x SIZE bytes from per-lane global addresses to LDS at (wave-uniform base + lane*SIZE).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_cvoid_t;
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gl_cvoid_t*)gsrc, (lds_void_t*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float* gsrc, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((gl_cvoid_t*)gsrc, (lds_void_t*)lds_dst, 4, 0, 0);
}
// wait until at most N vector-memory operations of this wave are outstanding, then the workgroup barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// explicit s_waitcnt vmcnt(0) that hipcc's waitcnt pass understands (expcnt/lgkmcnt left at max)
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// packed fp32 math: v_pk_fma_f32 does two FMAs per issue slot; with one wave per SIMD the kernels are
// issue-bound (one instruction per ~4 cycles per wave), so packing halves the cost of the FMA blocks.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// Element i of a tape that holds fp32 (bf = 0) or bf16 (bf = 1) values, with ONE unconditional load either way: a load
// under a condition is not issued before the one in front of it has returned (hipcc puts a vmcnt(0) there)
__device__ __forceinline__ float tape_elem(const float* base, unsigned i, int bf) {
  const unsigned w = reinterpret_cast<const unsigned*>(base)[bf ? (i >> 1) : i];
  return __uint_as_float(bf ? ((i & 1u) ? (w & 0xffff0000u) : (w << 16)) : w);
}
__device__ __forceinline__ unsigned short vg_f2bf(float f) {   // round to nearest even
  const unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Element of an array of argument structs inside the kernel-argument segment, at a run-time (wave-uniform) byte offset.
// The reference goes through a generic pointer derived from the segment pointer: the compiler infers the constant
// address space back and loads just the fields that are used, as scalar loads - like any by-value kernel argument.  (A
// dynamic index into the by-value argument itself is copied through scratch; loading the whole 600-byte struct into
// registers up front overflows the SGPR file: finish_stack_kernel took 13.9 us that way, 2 x 4.6 as two launches.)
template <typename S>
__device__ __forceinline__ const S& vg_karg_ref(size_t byte_offset) {
  const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  return *(const S*)(const char*)(ka + byte_offset);
}

// ---------------------------------------------------------------------------------------------------
// canonical element <- reference layouts (oracle/vmlmf_oracle.py: canonicalize)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int vg_hchunk(const VGeo& g, int k) { return g.hperm ? (k ^ 1) : k; }
__device__ __forceinline__ int vg_xchunk(const VGeo& g, int k) { return g.xperm ? (k ^ 1) : k; }
// per-gate pointer picked without indexing the kernel-argument array dynamically
template <class T>
__device__ __forceinline__ T* vg_gate(T* const (&a)[4], int k) { return k == 0 ? a[0] : k == 1 ? a[1] : k == 2 ? a[2] : a[3]; }

// The x-side accessors are branch-free (clamped index, masked value): finish_kernel calls them in short loops, and a load
// under a condition is not issued before the previous one has returned (hipcc puts a vmcnt(0) in front of it).
__device__ inline float ref_ux(const VGeo& g, const RefP& p, int m, int r) {
  const bool in = r < g.rw;
  const float v = p.u_x[(size_t)m * g.rw + (in ? r : 0)];
  return in ? v : 0.f;
}
__device__ inline float ref_vx(const VGeo& g, const RefP& p, int n, int k, int r) {
  const bool in = r < g.rw;
  const int rr = in ? r : 0;
  const float* base = g.pergate ? vg_gate(p.wg, k) : p.v_x;
  const size_t off = g.pergate ? (size_t)rr * g.H + n : ((size_t)vg_xchunk(g, k) * g.H + n) * g.rw + rr;
  const float v = base[off];
  return in ? v : 0.f;
}
// unit n's contribution weight to rank rr of the concatenated rank space
__device__ inline float ref_uc(const VGeo& g, const RefP& p, int n, int rr) {
  const int s = (g.G == 2 && rr >= g.off1) ? 1 : 0;
  const int r = rr - (s ? g.off1 : 0);
  const int rus = s ? g.ru1 : g.ru0;
  if (r >= rus) return 0.f;
  const float* u = s ? p.u_h1 : p.u_h0;
  if (g.G == 1) return u[(size_t)n * rus + r];
  const int grp = n / g.Hg, m = n - grp * g.Hg;
  const int j = (grp - s + g.G) % g.G;  // destination group of shift s
  return u[((size_t)j * g.Hg + m) * rus + r];
}
// (Q vector, column) that gate k of unit n reads in the (g, r, 4Hg) matrices
__device__ inline void vg_vc_loc(const VGeo& g, int n, int k, int& q, int& col) {
  if (g.flat) {
    const int f = k * g.H + n;
    q = f / (4 * g.Hg);
    col = f - q * 4 * g.Hg;
  } else {
    q = n / g.Hg;
    col = vg_hchunk(g, k) * g.Hg + (n - q * g.Hg);
  }
}
__device__ inline float ref_vc(const VGeo& g, const RefP& p, int n, int k, int rr) {
  const int s = (g.G == 2 && rr >= g.off1) ? 1 : 0;
  const int r = rr - (s ? g.off1 : 0);
  const int rus = s ? g.ru1 : g.ru0;
  if (r >= rus) return 0.f;
  if (g.pergate) return vg_gate(p.ug, k)[(size_t)r * g.H + n];
  const float* v = s ? p.v_h1 : p.v_h0;
  if (g.G == 1) return v[((size_t)k * g.H + n) * rus + r];
  int q, col;
  vg_vc_loc(g, n, k, q, col);
  return v[((size_t)q * rus + r) * (4 * g.Hg) + col];
}
__device__ inline float ref_bb(const VGeo& g, const RefP& p, int n, int k) {
  if (g.pergate) return vg_gate(p.bg, k)[n];
  return p.b_x[vg_xchunk(g, k) * g.H + n] + p.b_h[vg_hchunk(g, k) * g.H + n];
}
// element e of the partial blocks c .. c1 - 1 (PCH floats apart), summed in a FIXED order: four interleaved accumulators over batches
// of eight loads that are in flight together (reduce_cg_kernel, finish2_kernel: deterministic, no float atomics)
__device__ __forceinline__ float vg_block_sum(const float* __restrict__ Pall, const long long PCH, const long long e, int c, const int c1) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (; c < c1; c += 8) {   // (the last batch clamps its addresses and masks its values: as a loop of single loads the six blocks
                             //  behind 24 of K = 30 were six memory round trips in a row)
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = Pall[(size_t)(c + i < c1 ? c + i : c1 - 1) * PCH + e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i & 3] += c + i < c1 ? v[i] : 0.f;
  }
  return (s[0] + s[1]) + (s[2] + s[3]);
}

// thread slot -> unit
__device__ __forceinline__ bool vg_slot_unit(const VGeo& g, int slot, int& n) {
  const int grp = slot / (64 * g.W);
  const int m = slot - grp * 64 * g.W;
  n = grp * g.Hg + m;
  return m < g.Hg;
}
