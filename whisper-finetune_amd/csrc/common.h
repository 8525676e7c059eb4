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
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WFT_LDS __attribute__((address_space(3)))
#define WFT_GLB __attribute__((address_space(1)))

void wft_set_error(const char* fmt, ...);

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
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
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

// exact-erf GELU (torch.nn.GELU default / F.gelu)
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const WFT_GLB void*)gsrc, (WFT_LDS void*)lds_wave_base, 16, 0, 0);
}

// transposed LDS read: see cdna_hip_programming.md T10
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((WFT_LDS s16x4*)lds_addr);
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
