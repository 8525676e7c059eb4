// issue_rates.hip — developer micro-benchmark (GPU box): issue cost of the vector instructions the epilogues and the attention
// softmax are made of, per wave, with one wave per SIMD (the NT 4-wave kernel's situation) and with two.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_rates.hip -o tools/micro/issue_rates && tools/micro/issue_rates
// Each test is a loop of 16 x .rept 32 blocks of 8 instructions on independent registers (or one dependent chain); cycles by
// s_memtime around the loop, wave 0 of every workgroup, median over workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REPT_BODY(body) ".rept 32\n" body ".endr\n"

template <int T>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
  float b0 = 0.5f, b1 = 0.25f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {0.999f, 1.001f};
  typedef __attribute__((ext_vector_type(4))) float f4;
  typedef __attribute__((ext_vector_type(8))) short s8;
  f4 acc = {0, 0, 0, 0};
  s8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 2, 3, 4, 5, 6, 7, 8};
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  unsigned addr = ((threadIdx.x * 2654435761u) >> 20) & 0xff0;  // random 16-byte-aligned offsets inside 4 KiB
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; ++it) {
    if constexpr (T == 0) asm volatile(REPT_BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (T == 1) asm volatile(REPT_BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (T == 2) asm volatile(REPT_BODY("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    if constexpr (T == 3) asm volatile(REPT_BODY("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
    if constexpr (T == 4) asm volatile(REPT_BODY("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
    if constexpr (T == 5) asm volatile(REPT_BODY("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %1, %1, %0\n v_cvt_pk_bf16_f32 %3, %3, %2\n v_cvt_pk_bf16_f32 %5, %5, %4\n v_cvt_pk_bf16_f32 %7, %7, %6\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    // exp with 3 independent fma between consecutive exps: does the transcendental unit run beside the main VALU?
    if constexpr (T == 6) asm volatile(REPT_BODY("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    // exp immediately consumed (the hazard hipcc pads with s_nop)
    if constexpr (T == 7) asm volatile(REPT_BODY("v_exp_f32 %0, %0\n s_nop 0\n v_mul_f32 %0, %0, %8\n v_exp_f32 %1, %1\n s_nop 0\n v_mul_f32 %1, %1, %8\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %2, %2, %8\n v_exp_f32 %3, %3\n s_nop 0\n v_mul_f32 %3, %3, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    // LDS gather: ds_read_b128 at random 16-byte slots of a 4 KiB table (8 reads in flight)
    if constexpr (T == 8) {
      f4 r0, r1, r2, r3;
      asm volatile(REPT_BODY("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)\n") : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr));
      a0 += r0[0] + r1[1] + r2[2] + r3[3];
    }
    if constexpr (T == 9) asm volatile(REPT_BODY("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    // one MFMA 16x16x32 + 3 vector instructions per gap (1 exp + 2 fma): do they hide?
    if constexpr (T == 10) asm volatile(REPT_BODY("v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n v_exp_f32 %1, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(acc), "v"(fa), "v"(fb));
    if constexpr (T == 11) asm volatile(REPT_BODY("v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(acc), "v"(fa), "v"(fb));
    // one MFMA + 3 plain fma per gap
    if constexpr (T == 12) asm volatile(REPT_BODY("v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_mfma_f32_16x16x32_bf16 %10, %11, %12, %10\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(acc), "v"(fa), "v"(fb));
    // exp alternating with pk_fma (2 elements): the GELU mix
    if constexpr (T == 13) asm volatile(REPT_BODY("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %8, %8\n v_rcp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %8, %8\n v_exp_f32 %2, %2\n v_pk_fma_f32 %6, %6, %8, %8\n v_rcp_f32 %3, %3\n v_pk_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[0] + p2[1] + p3[1] + acc[0] == 12345.678f) out[0] = 0;
}

template <int T>
double run(int block, int per_iter) {
  unsigned long long* d;
  const int grid = 256;
  hipMalloc(&d, grid * 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<T>, dim3(grid), dim3(block), 0, 0, d, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  hipFree(d);
  std::sort(h.begin(), h.end());
  return (double)h[grid / 2] / (16.0 * 32.0 * per_iter);
}

int main() {
  const char* names[] = {"v_exp_f32 x8 independent", "v_rcp_f32 x8 independent", "v_fma_f32 x8 independent", "v_pk_fma_f32 x8 (4 regs)",
                         "v_pk_mul_f32 x8 (4 regs)", "v_cvt_pk_bf16_f32 x8", "exp + 3 fma (per group of 4)", "exp, s_nop 0, dependent mul (per triple)",
                         "ds_read_b128 random gather, 4 + wait (per read)", "v_accvgpr_read_b32 x8", "mfma16x16x32 + exp + 2 fma (per group)",
                         "mfma16x16x32 back to back (dependent chain)", "mfma16x16x32 + 3 fma (per group)", "exp|rcp alternating with pk_fma (per pair)"};
  const int per[] = {8, 8, 8, 8, 8, 8, 2, 4, 4, 8, 2, 2, 2, 4};
  printf("%-52s %12s %12s %12s  (s_memtime ticks per unit; the clock of s_memtime is 100 MHz on gfx950 if ticks look 20x too small)\n", "test", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
#define R(T) printf("%-52s %12.2f %12.2f %12.2f\n", names[T], run<T>(256, per[T]), run<T>(512, per[T]), run<T>(1024, per[T]));
  R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13)
  return 0;
}
