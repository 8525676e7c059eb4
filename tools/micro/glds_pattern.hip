// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) issue throughput for two per-instruction footprints of the same bytes:
//   mode 0: 16 rows x 64 B  (a 32-deep bf16 k-slab: each instruction touches 16 half cache lines)
//   mode 1:  8 rows x 128 B (a 64-deep k-slab: 8 full lines)
// 8 waves per workgroup, one workgroup per CU, every wave streams its share of a [rows][ld] bf16 matrix that stays in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(const unsigned short* a, long ld, int rows_per_block, int iters, int mode, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* base = a + (long)blockIdx.x * rows_per_block * ld;
  char* dst = lds + wave * 16384;
  for (int it = 0; it < iters; ++it) {
    const int kofs = (it * (mode ? 64 : 32)) % (int)ld;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned short* src;
      if (mode == 0) src = base + (long)(wave * 32 + (j & 1) * 16 + (lane >> 2)) * ld + kofs + (lane & 3) * 8 + (j >> 1) * 0;
      else src = base + (long)(wave * 32 + (j & 3) * 8 + (lane >> 3)) * ld + kofs + (lane & 7) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + (j & 15) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = ((float*)lds)[5];
}
int main() {
  const long ld = 1280; const int rpb = 256, nblk = 256, iters = 2000;
  unsigned short* a; float* sink;
  hipMalloc(&a, (size_t)nblk * rpb * ld * 2); hipMalloc(&sink, nblk * 4);
  hipMemset(a, 0, (size_t)nblk * rpb * ld * 2);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 131072, 0, a, ld, rpb, iters, mode, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)nblk * 8 * iters * 16 * 1024;
      printf("mode %d: %.3f ms  %.2f TB/s  (%.1f ns per wave-instruction per CU-wave)\n", mode, ms, bytes / ms / 1e9, ms * 1e6 / (iters * 16.0));
    }
  return 0;
}
