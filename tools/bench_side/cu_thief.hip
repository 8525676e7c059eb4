// cu_thief.hip — bench tooling, NOT part of libwft.so: a stand-in for RCCL's ring all-reduce kernel on a one-GPU box.
//
// bench.py's `ddp_mode_1gpu` block times the headline step the way a WORLD_SIZE = 8 job runs it — DDP
// wrapper, per-tile GEMM / attention launches — and once more while a kernel on a side stream HOLDS `n_wgs` compute units and
// moves the bytes the ring would move (2 (N-1)/N x the bucket, read and written) at the rate an xGMI ring sustains.  It is
// launched from a DDP communication hook, once per gradient bucket, exactly where torch's reducer would enqueue the
// all-reduce (the reference's DDP wrap: src/whisper_finetune/scripts/finetune.py:694-710).
//
// A workgroup copies 64 KiB chunks src -> dst (both inside caller-owned scratch buffers, wrapped) and paces itself on the
// 100 MHz wall clock so that the whole launch lasts move_bytes / rate: RCCL's kernels are link-bound, not HBM-bound — they sit
// on their CUs for the duration of the transfer, which is what takes tiles away from a persistent GEMM grid.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void cu_thief_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long buf_chunks,
                                                      long move_chunks, long ticks_total) {
  const long t0 = (long)wall_clock64();
  for (long c = blockIdx.x; c < move_chunks; c += gridDim.x) {
    const long due = t0 + (long)((double)ticks_total * (double)c / (double)move_chunks);
    while ((long)wall_clock64() < due) __builtin_amdgcn_s_sleep(32);
    const long base = (c % buf_chunks) * 4096;  // 4096 x 16 B = 64 KiB
#pragma unroll 4
    for (int i = threadIdx.x; i < 4096; i += 256) dst[base + i] = src[base + i];
  }
  while ((long)wall_clock64() < t0 + ticks_total) __builtin_amdgcn_s_sleep(32);  // hold the CU until the "transfer" ends
}

// src / dst: scratch buffers of buf_bytes each (multiples of 64 KiB); move_bytes: bytes to copy in this launch;
// rate_gbps: pacing (bytes moved per second / 1e9).  Returns 0 or the hipError_t of the launch.
extern "C" int wft_bench_cu_thief(const void* src, void* dst, long buf_bytes, long move_bytes, int n_wgs, double rate_gbps, void* stream) {
  const long buf_chunks = buf_bytes >> 16, move_chunks = (move_bytes + 65535) >> 16;
  if (buf_chunks <= 0 || move_chunks <= 0 || n_wgs <= 0 || rate_gbps <= 0) return -1;
  const long ticks = (long)((double)move_bytes / (rate_gbps * 1e9) * 1e8);  // wall_clock64: 100 MHz
  hipLaunchKernelGGL(cu_thief_kernel, dim3((unsigned)n_wgs), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                     buf_chunks, move_chunks, ticks);
  return (int)hipGetLastError();
}
