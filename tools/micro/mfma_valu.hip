// mfma_valu.hip — developer micro-benchmark (GPU box): how much vector work hides behind v_mfma_f32_32x32x16_bf16 (the attention
// kernels' MFMA; 8 passes = 32 cycles of the matrix pipe), per SIMD, with 1 and 2 waves per SIMD.  Every wave runs a loop of
// groups {2 MFMA on two independent accumulators + K vector instructions}; reported: cycles per group for the WHOLE workgroup
// (last wave's end - first wave's start, s_memtime), i.e. the SIMD's throughput, not one wave's view (the arbiter is oldest-first).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu.hip -o tools/micro/mfma_valu && tools/micro/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) short s8;

template <int K, int EXP, int M16>
__global__ __launch_bounds__(512) void k(unsigned long long* t_begin, unsigned long long* t_end, float seed) {
  f16v a0 = {0}, a1 = {0};
  s8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 2, 3, 4, 5, 6, 7, 8};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x + i;
  float b0 = 0.5f, b1 = 0.25f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (M16) {
        // the same 64 matrix-pipe cycles as FOUR 16x16x32 instructions (four independent accumulators), the vector work in four parts
        typedef __attribute__((ext_vector_type(4))) float f4v;
        f4v* q = (f4v*)&a0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q[m]) : "v"(fa), "v"(fb));
#pragma unroll
          for (int j = m * K / 4; j < (m + 1) * K / 4; ++j) {
            if (EXP && (j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b0), "v"(b1));
          }
        }
      } else {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(a0), "+v"(a1) : "v"(fa), "v"(fb));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (EXP && (j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b0), "v"(b1));
      }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) {
    t_begin[blockIdx.x * 8 + (threadIdx.x >> 6)] = t0;
    t_end[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1;
  }
  float s = a0[0] + a1[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) t_end[0] = 0;
}

template <int K, int EXP, int M16 = 0>
void run(int block) {
  const int grid = 256;
  unsigned long long *b, *e;
  hipMalloc(&b, grid * 8 * 8); hipMalloc(&e, grid * 8 * 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<K, EXP, M16>), dim3(grid), dim3(block), 0, 0, b, e, 1.0f);
  hipDeviceSynchronize();
  const int nw = block / 64;
  std::vector<unsigned long long> hb(grid * 8), he(grid * 8);
  hipMemcpy(hb.data(), b, grid * 64, hipMemcpyDeviceToHost); hipMemcpy(he.data(), e, grid * 64, hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int g = 0; g < grid; ++g) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; ++w) { lo = std::min(lo, hb[g * 8 + w]); hi = std::max(hi, he[g * 8 + w]); }
    per.push_back((double)(hi - lo) / (64.0 * 8.0));
  }
  std::sort(per.begin(), per.end());
  const int waves_per_simd = nw / 4;
  // a group = 2 MFMA (64 matrix-pipe cycles) + K vector instructions, per wave; per SIMD a "round" of groups = waves_per_simd groups
  printf("%s K=%2d vector (%s) per 64 pipe cycles, %d wave(s)/SIMD: %7.1f cycles per group-round = %6.1f per group per wave-slot; matrix pipe needs %d\n", M16 ? "4 x 16x16x32" : "2 x 32x32x16", K,
         EXP ? "1 exp in 4" : "fma", waves_per_simd, per[grid / 2], per[grid / 2] / waves_per_simd, 64 * waves_per_simd);
  hipFree(b); hipFree(e);
}
int main() {
#define R(K, E) run<K, E>(256); run<K, E>(512);
#define R16(K, E) run<K, E, 1>(256); run<K, E, 1>(512);
  R16(0, 0) R16(4, 0) R16(8, 0) R16(12, 0) R16(16, 0) R16(24, 0) R16(8, 1) R16(16, 1) R16(24, 1)
  R(0, 0) R(2, 0) R(4, 0) R(6, 0) R(8, 0) R(12, 0) R(16, 0) R(24, 0) R(8, 1) R(16, 1) R(24, 1) R(34, 1)
  return 0;
}
