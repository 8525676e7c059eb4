// common.h — shared device helpers for libwft (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wft.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WFT_LDS __attribute__((address_space(3)))
#define WFT_GLB __attribute__((address_space(1)))

void wft_set_error(const char* fmt, ...);

// Developer switches (WFT_GEMM_DIAG and the A/B variables listed in DESIGN.md §3) exist only in a library built with
// -DWFT_TIMING_BUILDS (`make TIMING=1` -> libwft_timing.so): some of them drop stores or epilogues "for timing only", and a stray
// variable in a job script must not be able to make the shipped library train on garbage.  The default build never reads them
// (the names are not even in the binary: tests/test_abi.py), and wft_version() says which build is loaded.
#include <stdlib.h>
#ifdef WFT_TIMING_BUILDS
static inline const char* wft_dev_getenv(const char* name) { return getenv(name); }
#define WFT_BUILD_KIND " timing-builds"
#else
static inline const char* wft_dev_getenv(const char*) { return nullptr; }
#define WFT_BUILD_KIND ""
#endif

#define WFT_CHECK_ARG(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) {                                                 \
      wft_set_error("%s: %s (%s)", __func__, msg, #cond);          \
      return WFT_ERR_ARG;                                          \
    }                                                              \
  } while (0)

#define WFT_CHECK_LAUNCH()                                         \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) {                                        \
      wft_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return WFT_ERR_LAUNCH;                                       \
    }                                                              \
  } while (0)

// ---- bf16 <-> f32 (round to nearest even; plain casts so hipcc emits v_cvt_pk_bf16_f32)
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
// two floats -> packed bf16 pair as ONE v_cvt_pk_bf16_f32 (the two scalar casts + shift + or compile to four instructions)
typedef __attribute__((ext_vector_type(2))) float wft_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 wft_bf16x2_t;
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  const wft_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, wft_bf16x2_t));
}

// ---- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf with |abs err| < 1.5e-7 (Abramowitz & Stegun 7.1.26): 1 rcp + 1 exp + 6 fma instead of
// ocml's erff (~3x the VALU); far below the bf16 resolution of every consumer.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}
// exact-erf GELU (torch.nn.GELU default / F.gelu)
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}
// gelu'(x) = Phi(x) + x phi(x): the exp(-x^2/2) inside erf_fast(x/sqrt2) IS phi's exponential — one v_exp + one v_rcp
__device__ __forceinline__ float dgelu_f(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2) = 2^(-x^2 / (2 ln 2))
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * e;                 // 0.5 * (1 - erf(|x|/sqrt2))
  const float cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
  return fmaf(x, 0.39894228040143267794f * e, cdf);
}

// gelu(x) = x Phi(x) and gelu'(x) from the same exponential, reciprocal and polynomial (WFT_EPI_GELU_GRAD: the forward
// GEMM stores gelu' so that the backward-data GEMM's epilogue is a single multiply)
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& d) {
  // (p / sqrt 2 in one constant; the polynomial's coefficients carry the factor 0.5 of half_erfc — exact, a power of two;
  // x^2 / (2 ln 2) as the product of two scaled copies of x: all three keep the packed-fp32 forms and save single-lane issues)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));
  const float xs = x * 0.84932180f;  // sqrt(1 / (2 ln 2))
  const float e = __builtin_amdgcn_exp2f(-(xs * xs));
  float poly = fmaf(0.5307027145f, t, -0.7265760135f);
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  const float half_erfc = poly * t * e;
  const float cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
  g = x * cdf;
  d = fmaf(x, 0.39894228040143267794f * e, cdf);
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const WFT_GLB void*)gsrc, (WFT_LDS void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA with the address split the way the hardware takes it: scalar 64-bit base + per-lane 32-bit byte offset, LDS
// destination (wave-uniform byte address) in M0.  Inline asm because hipcc folds base + offset into a per-lane 64-bit
// pointer and then spends a v_lshl_add_u64 per piece per slab on it (vector instructions next to a partner wave that
// issues MFMAs at raised priority are the expensive part of a load phase).  Counts in vmcnt like the builtin.  M0 is a
// reserved register for hipcc (it re-materialises M0 in front of every instruction of its own that reads it), so writing it
// here needs no clobber — listing it only draws -Winline-asm.
__device__ __forceinline__ void glds16_saddr(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// transposed LDS read: see cdna_hip_programming.md T10
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((WFT_LDS s16x4*)lds_addr);
}

// Same read issued from inline asm: hipcc (ROCm 7.2) cannot disambiguate the builtin from pending LDS-DMA
// writes and drains them with s_waitcnt vmcnt(0) in front of every read.  The caller owns the wait:
// asm volatile("s_waitcnt lgkmcnt(0)") followed by __builtin_amdgcn_sched_barrier(0) before the first use
// (cdna_hip_programming.md §5.7 item 1 form (iii), rule 18).
__device__ __forceinline__ s16x4 lds_read_tr16_asm(unsigned lds_byte_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
  return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const WFT_LDS char*)p;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
