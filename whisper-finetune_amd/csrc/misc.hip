// misc.hip — HBM-bound side kernels of the training step: casts / padded transposed weight
// shadows, residual-gradient add, bias-gradient column sums, token embedding fwd/bwd,
// label-smoothed cross entropy fwd/bwd (+ teacher-forced argmax), AdamW, sum of squares.
// All use 16-byte per-lane accesses where the layout allows (cdna_hip_programming.md G13).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void wft_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* wft_last_error(void) { return g_err; }
extern "C" const char* wft_version(void) { return "wft 0.1 gfx950" WFT_BUILD_KIND; }

static inline int ew_grid(int64_t nvec) {
  int64_t g = (nvec + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ----------------------------------------------------------------------------- casts
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* src, unsigned short* dst, long n) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const f32x4 a = *(const f32x4*)(src + i * 8), b = *(const f32x4*)(src + i * 8 + 4);
    u32x4 o = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    *(u32x4*)(dst + i * 8) = o;
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 3) + threadIdx.x;
    if (t < n) dst[t] = f2bf(src[t]);
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const unsigned short* src, float* dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = bf2f(src[i]);
}

extern "C" int wft_cast_f32_bf16(const float* src, wft_bf16* dst, int64_t n, void* stream) {
  WFT_CHECK_ARG(src && dst && n >= 0, "bad args");
  WFT_CHECK_ARG((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, "16-byte alignment");
  if (n == 0) return WFT_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
extern "C" int wft_cast_bf16_f32(const wft_bf16* src, float* dst, int64_t n, void* stream) {
  WFT_CHECK_ARG(src && dst && n >= 0, "bad args");
  if (n == 0) return WFT_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// src f32 [rows, cols] -> dst bf16 [rows_pad, cols_pad], dst_t bf16 [cols_pad, rows_pad]; 64x64 tiles via LDS.
// Each thread moves 4 consecutive elements (16-B loads, 8-B stores) on both the straight and the transposed side;
// `fast` = every pointer / leading dimension allows that (checked on the host), otherwise element-wise.
__global__ __launch_bounds__(256) void cast_pad_t_kernel(const float* src, long rows, long cols, unsigned short* dst,
                                                          unsigned short* dst_t, long rows_pad, long cols_pad,
                                                          long ld_dst, long ld_dst_t, int fast, float fs) {
  // fs (fwd_scale): dst = bf16(fs * src) — ONE rounding of the scaled value — while dst_t stays bf16(src): the softmax scale folded
  // into the forward shadow of an attention q projection (wft_attn_args.q_prescaled)
  __shared__ unsigned short tile[64][68];
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  if (fast) {
    const int q = threadIdx.x & 15, rr0 = threadIdx.x >> 4;  // 16 threads x 4 columns per row, 16 rows per pass
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 16 + rr0;
      const long r = r0 + rr, c = c0 + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && c + 3 < cols) v = *(const f32x4*)(src + r * cols + c);
      else if (r < rows) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c + e < cols) ? src[r * cols + c + e] : 0.f;
      }
      const u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *(u32x2*)&tile[rr][q * 4] = pk;
      const u32x2 pf = {pack2bf(v[0] * fs, v[1] * fs), pack2bf(v[2] * fs, v[3] * fs)};
      if (r < rows_pad && c < cols_pad) *(u32x2*)(dst + r * ld_dst + c) = pf;  // cols_pad % 4 == 0 in fast mode
    }
    if (dst_t) {
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int cc = pass * 16 + rr0;  // transposed row = source column
        const long c = c0 + cc, r = r0 + q * 4;
        if (c < cols_pad && r < rows_pad) {
          const u32x2 pk = {(unsigned)tile[q * 4][cc] | ((unsigned)tile[q * 4 + 1][cc] << 16),
                            (unsigned)tile[q * 4 + 2][cc] | ((unsigned)tile[q * 4 + 3][cc] << 16)};
          *(u32x2*)(dst_t + c * ld_dst_t + r) = pk;  // rows_pad % 4 == 0 in fast mode
        }
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const long r = r0 + rr, c = c0 + tx;
    unsigned short v = 0, vf = 0;
    if (r < rows && c < cols) { v = f2bf(src[r * cols + c]); vf = f2bf(src[r * cols + c] * fs); }
    tile[rr][tx] = v;
    if (r < rows_pad && c < cols_pad) dst[r * ld_dst + c] = vf;
  }
  if (dst_t) {
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
      const long c = c0 + cc, r = r0 + tx;
      if (c < cols_pad && r < rows_pad) dst_t[c * ld_dst_t + r] = tile[tx][cc];
    }
  }
}
extern "C" int wft_cast_pad_transpose_f32_bf16(const float* src, int64_t rows, int64_t cols, wft_bf16* dst,
                                               wft_bf16* dst_t, int64_t rows_pad, int64_t cols_pad, int64_t ld_dst,
                                               int64_t ld_dst_t, float fwd_scale, void* stream) {
  WFT_CHECK_ARG(src && dst, "null pointer");
  if (fwd_scale == 0.f) fwd_scale = 1.f;
  WFT_CHECK_ARG(rows >= 1 && cols >= 1 && rows_pad >= rows && cols_pad >= cols, "bad shape");
  WFT_CHECK_ARG(ld_dst >= cols_pad && (!dst_t || ld_dst_t >= rows_pad), "leading dimensions too small");
  dim3 grid((unsigned)((cols_pad + 63) / 64), (unsigned)((rows_pad + 63) / 64));
  const int fast = cols % 4 == 0 && cols_pad % 4 == 0 && rows_pad % 4 == 0 && ld_dst % 4 == 0 && (!dst_t || ld_dst_t % 4 == 0) &&
                   (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0 && (!dst_t || (((uintptr_t)dst_t) & 7) == 0);
  hipLaunchKernelGGL(cast_pad_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (long)rows, (long)cols, dst, dst_t,
                     (long)rows_pad, (long)cols_pad, (long)ld_dst, (long)ld_dst_t, fast, fwd_scale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// W_eff = W + scaling * B (A ⊙ mask): minLoRA's parametrized weight (merge: model/lora.py:83-89).
// Written as bf16 [rows_pad, cols_pad] (+ transposed) for the GEMMs and/or as f32 [rows, cols] (merge_lora; may alias W).
// 64x64 tiles; the tile's B rows and (masked, scaled) A columns sit in LDS: r <= 64 FMAs per element, HBM-bound.
__global__ __launch_bounds__(256) void lora_merge_kernel(const float* W, long rows, long cols, const float* Bm, const float* Am,
                                                          const float* mask, int r, float scaling, unsigned short* dst,
                                                          unsigned short* dst_t, long rows_pad, long cols_pad, long ld_dst,
                                                          long ld_dst_t, float* dst_f32, int fast, float fs) {
  __shared__ unsigned short tile[64][68];
  __shared__ float bs[64][65];                                  // bs[i][q] = B[r0 + i][q]
  __shared__ __attribute__((aligned(16))) float as[64][68];     // as[q][j] = scaling * A[q][c0 + j] * mask[c0 + j]
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * r; i += 256) {
    const int a = i / r, q = i - a * r;
    bs[a][q] = (r0 + a < rows) ? Bm[(r0 + a) * r + q] : 0.f;
  }
  for (int i = threadIdx.x; i < 64 * r; i += 256) {
    const int q = i >> 6, j = i & 63;
    const long c = c0 + j;
    as[q][j] = (c < cols) ? scaling * Am[(long)q * cols + c] * (mask ? mask[c] : 1.f) : 0.f;
  }
  __syncthreads();
  if (fast) {  // 4 consecutive columns per thread: 16-B loads of W / A, 8-B stores both ways (cf. cast_pad_t_kernel)
    const int q4 = threadIdx.x & 15, rr0 = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 16 + rr0;
      const long rw = r0 + rr, c = c0 + q4 * 4;
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      const bool in = rw < rows && c + 3 < cols;
      if (in) {
        w = *(const f32x4*)(W + rw * cols + c);
        for (int q = 0; q < r; ++q) {
          const float bq = bs[rr][q];
          const f32x4 a4 = *(const f32x4*)&as[q][q4 * 4];
          w[0] = fmaf(bq, a4[0], w[0]); w[1] = fmaf(bq, a4[1], w[1]); w[2] = fmaf(bq, a4[2], w[2]); w[3] = fmaf(bq, a4[3], w[3]);
        }
        if (dst_f32) *(f32x4*)(dst_f32 + rw * cols + c) = w;
      } else if (rw < rows) {
        for (int e = 0; e < 4; ++e)
          if (c + e < cols) {
            float acc = W[rw * cols + c + e];
            for (int q = 0; q < r; ++q) acc = fmaf(bs[rr][q], as[q][q4 * 4 + e], acc);
            w[e] = acc;
            if (dst_f32) dst_f32[rw * cols + c + e] = acc;
          }
      }
      const u32x2 pk = {pack2bf(w[0], w[1]), pack2bf(w[2], w[3])};
      *(u32x2*)&tile[rr][q4 * 4] = pk;
      const u32x2 pf = {pack2bf(w[0] * fs, w[1] * fs), pack2bf(w[2] * fs, w[3] * fs)};  // (fs: see cast_pad_t_kernel)
      if (dst && rw < rows_pad && c < cols_pad) *(u32x2*)(dst + rw * ld_dst + c) = pf;
    }
    if (dst_t) {
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int cc = pass * 16 + rr0;
        const long c = c0 + cc, rw = r0 + q4 * 4;
        if (c < cols_pad && rw < rows_pad) {
          const u32x2 pk = {(unsigned)tile[q4 * 4][cc] | ((unsigned)tile[q4 * 4 + 1][cc] << 16),
                            (unsigned)tile[q4 * 4 + 2][cc] | ((unsigned)tile[q4 * 4 + 3][cc] << 16)};
          *(u32x2*)(dst_t + c * ld_dst_t + rw) = pk;
        }
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const long rw = r0 + rr, c = c0 + tx;
    unsigned short v = 0, vf = 0;
    if (rw < rows && c < cols) {
      float acc = W[rw * cols + c];
      for (int q = 0; q < r; ++q) acc = fmaf(bs[rr][q], as[q][tx], acc);
      if (dst_f32) dst_f32[rw * cols + c] = acc;
      v = f2bf(acc);
      vf = f2bf(acc * fs);
    }
    tile[rr][tx] = v;
    if (dst && rw < rows_pad && c < cols_pad) dst[rw * ld_dst + c] = vf;
  }
  if (dst_t) {
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
      const long c = c0 + cc, rw = r0 + tx;
      if (c < cols_pad && rw < rows_pad) dst_t[c * ld_dst_t + rw] = tile[tx][cc];
    }
  }
}
extern "C" int wft_lora_merge(const float* W, int64_t rows, int64_t cols, const float* B, const float* A, const float* mask,
                              int rank, float scaling, wft_bf16* dst, wft_bf16* dst_t, int64_t rows_pad, int64_t cols_pad,
                              int64_t ld_dst, int64_t ld_dst_t, float* dst_f32, float fwd_scale, void* stream) {
  WFT_CHECK_ARG(W && B && A && (dst || dst_f32), "null pointer");
  if (fwd_scale == 0.f) fwd_scale = 1.f;
  WFT_CHECK_ARG(rows >= 1 && cols >= 1 && rank >= 1 && rank <= 64, "rank must be in 1..64");
  WFT_CHECK_ARG(!dst || (rows_pad >= rows && cols_pad >= cols && ld_dst >= cols_pad), "bad bf16 destination shape");
  WFT_CHECK_ARG(!dst_t || (dst && ld_dst_t >= rows_pad), "transposed destination needs dst and ld_dst_t >= rows_pad");
  if (!dst) { rows_pad = rows; cols_pad = cols; }
  dim3 grid((unsigned)((cols_pad + 63) / 64), (unsigned)((rows_pad + 63) / 64));
  const int fast = cols % 4 == 0 && cols_pad % 4 == 0 && rows_pad % 4 == 0 && (!dst || ld_dst % 4 == 0) && (!dst_t || ld_dst_t % 4 == 0) &&
                   (((uintptr_t)W) & 15) == 0 && (!dst || (((uintptr_t)dst) & 7) == 0) && (!dst_t || (((uintptr_t)dst_t) & 7) == 0) &&
                   (!dst_f32 || (((uintptr_t)dst_f32) & 15) == 0);
  hipLaunchKernelGGL(lora_merge_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, (long)rows, (long)cols, B, A, mask, rank,
                     scaling, dst, dst_t, (long)rows_pad, (long)cols_pad, (long)ld_dst, (long)ld_dst_t, dst_f32, fast, fwd_scale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// Operands of the rank-r adapter-gradient GEMMs of ONE adapter inside its Linear group's padded buffers (zero-initialised by
// the caller once; only this adapter's blocks are written):
//   Am [Rpad, K]  rows ro..ro+r  = bf16(scaling * A * mask)      AmT [K, Rpad] its transpose        (u = x Am^T carries s)
//   Bb [Npad, Rpad] block (no..no+n, ro..ro+r) = bf16(scaling * B)   BbT [Rpad, Npad] its transpose (du = dy Bb carries s)
// so dA = (du^T x) * mask and dB = dy^T u need no further scaling.  One launch instead of two element-wise multiplies and two
// cast/transposes per group; the data is a few tens of KB.
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* A, const float* mask, const float* B, int r, long K, long n,
                                                         float scaling, unsigned short* Am, unsigned short* AmT, unsigned short* Bb,
                                                         unsigned short* BbT, long rpad, long npad, long ro, long no) {
  const long na = (long)r * K, nb = n * r;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long)gridDim.x * 256) {
    if (i < na) {
      const long q = i / K, c = i - q * K;
      const unsigned short v = f2bf(scaling * A[i] * (mask ? mask[c] : 1.f));
      Am[(ro + q) * K + c] = v;
      AmT[c * rpad + ro + q] = v;
    } else {
      const long j = i - na, row = j / r, q = j - row * r;
      const unsigned short v = f2bf(scaling * B[j]);
      Bb[(no + row) * rpad + ro + q] = v;
      BbT[(ro + q) * npad + no + row] = v;
    }
  }
}
extern "C" int wft_lora_pack(const float* A, const float* mask, const float* B, int rank, int64_t K, int64_t n, float scaling,
                             wft_bf16* Am, wft_bf16* AmT, wft_bf16* Bb, wft_bf16* BbT, int64_t rpad, int64_t npad, int64_t ro,
                             int64_t no, void* stream) {
  WFT_CHECK_ARG(A && B && Am && AmT && Bb && BbT, "null pointer");
  WFT_CHECK_ARG(rank >= 1 && K >= 1 && n >= 1 && ro >= 0 && no >= 0 && ro + rank <= rpad && no + n <= npad, "bad shape");
  const int64_t total = (int64_t)rank * K + n * rank;
  hipLaunchKernelGGL(lora_pack_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, A, mask, B, rank, (long)K, (long)n,
                     scaling, Am, AmT, Bb, BbT, (long)rpad, (long)npad, (long)ro, (long)no);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// All adapters of a model in ONE launch: wft_lora_merge (bf16 shadow + transposed shadow) and wft_lora_pack for every row of
// the table — what the 2 x 512 per-Linear launches of a large-v3 LoRA forward/backward do, once per forward, right after the
// dropout masks are drawn.  Row layout (int64 x WFT_LORA_MT_FIELDS): see wft.h.  Blocks are 64x64 tiles of the weights, found by
// bisection over tile_start; the tile in the first row band also writes its 64 columns of Am / AmT, the tile in the first column
// band its 64 rows of Bb / BbT (the values are already in LDS for the merge).  Every value is computed exactly as by the
// per-adapter kernels.
#define WFT_LORA_MT_FIELDS 20
__global__ __launch_bounds__(256) void lora_refresh_mt_kernel(const long* tab, const int* tile_start, int n) {
  __shared__ unsigned short tile[64][68];
  __shared__ float bs[64][65];
  __shared__ __attribute__((aligned(16))) float as[64][68];
  int lo = 0, hi = n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const long* e = tab + (long)lo * WFT_LORA_MT_FIELDS;
  const float* W = (const float*)e[0];
  const long rows = e[1], cols = e[2];
  const float* Bm = (const float*)e[3];
  const float* Am = (const float*)e[4];
  const float* mask = (const float*)e[5];
  const int r = (int)e[6];
  const float scaling = __int_as_float((int)(e[7] & 0xffffffffL));
  const int fsb = (int)(e[7] >> 32);  // bits 32..63: fwd_scale as f32 bits (0 = 1.0): dst = bf16(fs * w), dst_t = bf16(w) — see cast_pad_t_kernel
  const float fs = fsb ? __int_as_float(fsb) : 1.f;
  unsigned short* dst = (unsigned short*)e[8];
  unsigned short* dst_t = (unsigned short*)e[9];
  const long ld_dst = e[10], ld_dst_t = e[11];
  unsigned short* pAm = (unsigned short*)e[12];
  unsigned short* pAmT = (unsigned short*)e[13];
  unsigned short* pBb = (unsigned short*)e[14];
  unsigned short* pBbT = (unsigned short*)e[15];
  const long rpad = e[16], npad = e[17], ro = e[18], no = e[19];
  const int t = bid - tile_start[lo];
  const int tiles_x = (int)(cols >> 6);
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const long r0 = (long)ty * 64, c0 = (long)tx * 64;
  for (int i = threadIdx.x; i < 64 * r; i += 256) {
    const int a = i / r, q = i - a * r;
    bs[a][q] = Bm[(r0 + a) * r + q];
  }
  for (int i = threadIdx.x; i < 64 * r; i += 256) {
    const int q = i >> 6, j = i & 63;
    const long c = c0 + j;
    as[q][j] = scaling * Am[(long)q * cols + c] * (mask ? mask[c] : 1.f);
  }
  __syncthreads();
  if (pAm && ty == 0)
    for (int i = threadIdx.x; i < 64 * r; i += 256) {
      const int q = i >> 6, j = i & 63;
      const unsigned short v = f2bf(as[q][j]);
      pAm[(ro + q) * cols + c0 + j] = v;
      pAmT[(c0 + j) * rpad + ro + q] = v;
    }
  if (pBb && tx == 0)
    for (int i = threadIdx.x; i < 64 * r; i += 256) {
      const int a = i / r, q = i - a * r;
      const unsigned short v = f2bf(scaling * bs[a][q]);
      pBb[(no + r0 + a) * rpad + ro + q] = v;
      pBbT[(ro + q) * npad + no + r0 + a] = v;
    }
  const int q4 = threadIdx.x & 15, rr0 = threadIdx.x >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int rr = pass * 16 + rr0;
    const long rw = r0 + rr, c = c0 + q4 * 4;
    f32x4 w = *(const f32x4*)(W + rw * cols + c);
    for (int q = 0; q < r; ++q) {
      const float bq = bs[rr][q];
      const f32x4 a4 = *(const f32x4*)&as[q][q4 * 4];
      w[0] = fmaf(bq, a4[0], w[0]); w[1] = fmaf(bq, a4[1], w[1]); w[2] = fmaf(bq, a4[2], w[2]); w[3] = fmaf(bq, a4[3], w[3]);
    }
    const u32x2 pk = {pack2bf(w[0], w[1]), pack2bf(w[2], w[3])};
    *(u32x2*)&tile[rr][q4 * 4] = pk;
    const u32x2 pf = {pack2bf(w[0] * fs, w[1] * fs), pack2bf(w[2] * fs, w[3] * fs)};
    *(u32x2*)(dst + rw * ld_dst + c) = pf;
  }
  if (dst_t) {
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int cc = pass * 16 + rr0;
      const long c = c0 + cc, rw = r0 + q4 * 4;
      const u32x2 pk = {(unsigned)tile[q4 * 4][cc] | ((unsigned)tile[q4 * 4 + 1][cc] << 16),
                        (unsigned)tile[q4 * 4 + 2][cc] | ((unsigned)tile[q4 * 4 + 3][cc] << 16)};
      *(u32x2*)(dst_t + c * ld_dst_t + rw) = pk;
    }
  }
}
extern "C" int wft_lora_refresh_mt(const void* tab, const int32_t* tile_start, int n, int total_tiles, void* stream) {
  WFT_CHECK_ARG(tab && tile_start && n >= 0 && total_tiles >= 0, "bad args");
  if (n == 0 || total_tiles == 0) return WFT_OK;
  hipLaunchKernelGGL(lora_refresh_mt_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const long*)tab,
                     (const int*)tile_start, n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

__global__ __launch_bounds__(256) void add_bf16_kernel(const unsigned short* a, const unsigned short* b,
                                                        unsigned short* y, long n) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const u32x4 x = *(const u32x4*)(a + i * 8), z = *(const u32x4*)(b + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2bf(bf2f((unsigned short)(x[e] & 0xffff)) + bf2f((unsigned short)(z[e] & 0xffff)),
                     bf2f((unsigned short)(x[e] >> 16)) + bf2f((unsigned short)(z[e] >> 16)));
    *(u32x4*)(y + i * 8) = o;
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 3) + threadIdx.x;
    if (t < n) y[t] = f2bf(bf2f(a[t]) + bf2f(b[t]));
  }
}
extern "C" int wft_add_bf16(const wft_bf16* a, const wft_bf16* b, wft_bf16* y, int64_t n, void* stream) {
  WFT_CHECK_ARG(a && b && y && n >= 0, "bad args");
  WFT_CHECK_ARG((((uintptr_t)a) & 15) == 0 && (((uintptr_t)b) & 15) == 0 && (((uintptr_t)y) & 15) == 0, "16-byte alignment");
  if (n == 0) return WFT_OK;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}


// ----------------------------------------------------------------------------- stochastic-depth rescale
// out = a*x + b*y (y may be NULL: out = a*x).  Forward of StochasticDepthMixin's train-time rescale
// x + (block(x) - x) / (1 - p) = (1 - s) x + s block(x), s = 1/(1-p)  (model/model_utils.py:241-250) in ONE pass instead
// of three element-wise kernels; its backward is two scaled copies.
__global__ __launch_bounds__(256) void axpby_bf16_kernel(float a, const unsigned short* x, float b, const unsigned short* y,
                                                          unsigned short* out, long n) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const u32x4 xv = *(const u32x4*)(x + i * 8);
    u32x4 yv = {0u, 0u, 0u, 0u};
    if (y) yv = *(const u32x4*)(y + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2bf(a * bf2f((unsigned short)(xv[e] & 0xffff)) + b * bf2f((unsigned short)(yv[e] & 0xffff)),
                     a * bf2f((unsigned short)(xv[e] >> 16)) + b * bf2f((unsigned short)(yv[e] >> 16)));
    *(u32x4*)(out + i * 8) = o;
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 3) + threadIdx.x;
    if (t < n) out[t] = f2bf(a * bf2f(x[t]) + (y ? b * bf2f(y[t]) : 0.f));
  }
}
extern "C" int wft_axpby_bf16(float a, const wft_bf16* x, float b, const wft_bf16* y, wft_bf16* out, int64_t n, void* stream) {
  WFT_CHECK_ARG(x && out && n >= 0, "bad args");
  WFT_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "16-byte alignment");
  if (n == 0) return WFT_OK;
  hipLaunchKernelGGL(axpby_bf16_kernel, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, a, x, b, y, out, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}


// ----------------------------------------------------------------------------- dGELU
// out = dy * gelu'(pre)   (conv stem backward; the Linear path fuses this in the GEMM epilogue)
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const unsigned short* dy, const unsigned short* pre,
                                                         unsigned short* out, long n) {
  const long nv = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const u32x4 a = *(const u32x4*)(dy + i * 8), b = *(const u32x4*)(pre + i * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2bf(bf2f((unsigned short)(a[e] & 0xffff)) * dgelu_f(bf2f((unsigned short)(b[e] & 0xffff))),
                     bf2f((unsigned short)(a[e] >> 16)) * dgelu_f(bf2f((unsigned short)(b[e] >> 16))));
    *(u32x4*)(out + i * 8) = o;
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 3) + threadIdx.x;
    if (t < n) out[t] = f2bf(bf2f(dy[t]) * dgelu_f(bf2f(pre[t])));
  }
}
extern "C" int wft_dgelu_mul_bf16(const wft_bf16* dy, const wft_bf16* pre, wft_bf16* out, int64_t n, void* stream) {
  WFT_CHECK_ARG(dy && pre && out && n >= 0, "bad args");
  WFT_CHECK_ARG((((uintptr_t)dy) & 15) == 0 && (((uintptr_t)pre) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "16-byte alignment");
  if (n == 0) return WFT_OK;
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, dy, pre, out, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ----------------------------------------------------------------------------- column sums
// out[c] (+)= sum_r x[r, c] without atomics (bitwise reproducible): a workgroup owns 32 columns for ALL rows — 4 column
// threads (16 bytes = 8 columns each, 64-byte row segments) x 64 row lanes — and folds its 64 partial rows in a fixed tree.
// Only reached where no producer kernel has formed the sums already (small models, the conv stem): the bf16 hot path gets
// its bias gradients from the LayerNorm-backward / GEMM / attention epilogues.
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* x, long rows, long cols, long ld, float* out,
                                                      int accumulate) {
  __shared__ float red[64][33];
  const int cx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const long c0 = (long)blockIdx.x * 32 + cx * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < cols) {
    for (long r = ry; r < rows; r += 64) {
      const u32x4 v = *(const u32x4*)(x + r * ld + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] += bf2f((unsigned short)(v[e] & 0xffff));
        s[2 * e + 1] += bf2f((unsigned short)(v[e] >> 16));
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] = s[e];
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (ry < o) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] += red[ry + o][cx * 8 + e];
    }
    __syncthreads();
  }
  if (threadIdx.x < 32) {
    const long c = (long)blockIdx.x * 32 + threadIdx.x;
    if (c < cols) out[c] = accumulate ? out[c] + red[0][threadIdx.x] : red[0][threadIdx.x];
  }
}
extern "C" int wft_colsum_bf16(const wft_bf16* x, int64_t rows, int64_t cols, int64_t ld, float* out, int accumulate,
                               void* stream) {
  WFT_CHECK_ARG(x && out, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && ld % 8 == 0, "cols/ld must be multiples of 8");
  WFT_CHECK_ARG((((uintptr_t)x) & 15) == 0, "16-byte alignment");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, (long)rows, (long)cols,
                     (long)ld, out, accumulate);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// Large inputs (the conv stem's bias gradients: 204 000 x 1280 at 68 clips): cols / 32 workgroups leave 216 of the 256 CUs idle
// (1.2 ms for 522 MB).  With a caller workspace the rows are cut into chunks, one workgroup per (column group, chunk) writes a
// partial row, and a second kernel adds the chunks in index order — the same fixed-order arithmetic at the HBM rate.
__global__ __launch_bounds__(256) void colsum_chunk_kernel(const unsigned short* x, long rows, long cols, long ld, long per, float* part) {
  __shared__ float red[64][33];
  const int cx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const long c0 = (long)blockIdx.x * 32 + cx * 8;
  const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < cols) {
    for (long r = r0 + ry; r < r1; r += 64) {
      const u32x4 v = *(const u32x4*)(x + r * ld + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] += bf2f((unsigned short)(v[e] & 0xffff));
        s[2 * e + 1] += bf2f((unsigned short)(v[e] >> 16));
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] = s[e];
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (ry < o) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] += red[ry + o][cx * 8 + e];
    }
    __syncthreads();
  }
  if (threadIdx.x < 32) {
    const long c = (long)blockIdx.x * 32 + threadIdx.x;
    if (c < cols) part[(long)blockIdx.y * cols + c] = red[0][threadIdx.x];
  }
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* part, int nchunk, long cols, float* out, int accumulate) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float t = accumulate ? out[c] : 0.f;
  for (int k = 0; k < nchunk; ++k) t += part[(long)k * cols + c];
  out[c] = t;
}
#define WFT_COLSUM_CHUNKS 64
extern "C" int64_t wft_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  // (round 6: from 8 192 rows on, was 65 536 — the conv stem's bias gradients of a whisper-base step, 24 000 x 512, took 94 us on the
  // 16 workgroups of the one-pass kernel)
  return rows >= 8192 ? (int64_t)WFT_COLSUM_CHUNKS * cols * (int64_t)sizeof(float) : 0;
}
extern "C" int wft_colsum_bf16_ws(const wft_bf16* x, int64_t rows, int64_t cols, int64_t ld, float* out, int accumulate,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  WFT_CHECK_ARG(x && out, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && ld % 8 == 0, "cols/ld must be multiples of 8");
  WFT_CHECK_ARG((((uintptr_t)x) & 15) == 0, "16-byte alignment");
  const int64_t need = wft_colsum_workspace_bytes(rows, cols);
  if (need == 0 || !workspace || workspace_bytes < need) return wft_colsum_bf16(x, rows, cols, ld, out, accumulate, stream);
  long per = (rows + WFT_COLSUM_CHUNKS - 1) / WFT_COLSUM_CHUNKS;
  if (per < 256) per = 256;  // (four passes of a workgroup's 64 row lanes at least)
  const int nchunk = (int)((rows + per - 1) / per);
  hipLaunchKernelGGL(colsum_chunk_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, x,
                     (long)rows, (long)cols, (long)ld, per, (float*)workspace);
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nchunk, (long)cols, out, accumulate);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ----------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long* tokens, const float* emb, const float* pos,
                                                         unsigned short* out, long n_tok, long S, int d, long V) {
  const int dv = d >> 3;
  const long total = n_tok * dv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long t = i / dv;
    const int c = (int)(i - t * dv) * 8;
    long tok = tokens[t];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const float* e = emb + tok * d + c;
    const float* pp = pos + (t % S) * d + c;
    const f32x4 a0 = *(const f32x4*)e, a1 = *(const f32x4*)(e + 4);
    const f32x4 b0 = *(const f32x4*)pp, b1 = *(const f32x4*)(pp + 4);
    u32x4 o = {pack2bf(a0[0] + b0[0], a0[1] + b0[1]), pack2bf(a0[2] + b0[2], a0[3] + b0[3]),
               pack2bf(a1[0] + b1[0], a1[1] + b1[1]), pack2bf(a1[2] + b1[2], a1[3] + b1[3])};
    *(u32x4*)(out + t * d + c) = o;
  }
}
extern "C" int wft_embed_fwd(const int64_t* tokens, const float* emb, const float* pos, wft_bf16* out, int64_t B,
                             int64_t S, int d, int64_t V, void* stream) {
  WFT_CHECK_ARG(tokens && emb && pos && out, "null pointer");
  WFT_CHECK_ARG(B >= 1 && S >= 1 && d >= 8 && d % 8 == 0 && V >= 1, "bad shape");
  const long total = B * S * (d / 8);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const long*)tokens, emb,
                     pos, out, (long)(B * S), (long)S, d, (long)V);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// demb[tok] += dout[position] for every position holding `tok`, WITHOUT atomics (a token id usually occurs at several
// positions — the special tokens at every clip's start — and fp32 atomics would add them in a run-dependent order): a
// workgroup owns 16 consecutive vocabulary rows, scans the token list in chunks of 256 and adds the positions that fall
// into its rows in position order (read-modify-write of rows nobody else touches).  ~V/16 blocks x n_tok/256 chunk scans of
// an L2-resident list.
__global__ __launch_bounds__(256) void embed_bwd_tok_kernel(const long* tokens, const unsigned short* dout, float* demb,
                                                             long n_tok, int d, long V) {
  __shared__ int hit[256];
  const long row0 = (long)blockIdx.x * 16;
  for (long base = 0; base < n_tok; base += 256) {
    const long j = base + threadIdx.x;
    const long tok = j < n_tok ? tokens[j] : -1;
    const bool mine = tok >= row0 && tok < row0 + 16 && tok < V;
    if (!__syncthreads_or(mine)) continue;  // block-uniform
    hit[threadIdx.x] = mine ? (int)(tok - row0) : -1;
    __syncthreads();
    for (int t = 0; t < 256; ++t) {
      const int r = hit[t];  // LDS broadcast: uniform
      if (r < 0) continue;
      const unsigned short* src = dout + (base + t) * d;
      float* dst = demb + (row0 + r) * d;
      for (int c = threadIdx.x; c < d; c += 256) dst[c] += bf2f(src[c]);
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const unsigned short* dout, float* dpos, long B, long S, int d) {
  const long total = S * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (long b = 0; b < B; ++b) s += bf2f(dout[b * S * d + i]);
    dpos[i] += s;
  }
}
extern "C" int wft_embed_bwd(const int64_t* tokens, const wft_bf16* dout, float* demb, float* dpos, int64_t B, int64_t S,
                             int d, int64_t V, void* stream) {
  WFT_CHECK_ARG(tokens && dout && demb && dpos, "null pointer");
  WFT_CHECK_ARG(B >= 1 && S >= 1 && d >= 1 && V >= 1, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3((unsigned)((V + 15) / 16)), dim3(256), 0, s, (const long*)tokens, dout, demb,
                     (long)(B * S), d, (long)V);
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(ew_grid(S * d)), dim3(256), 0, s, dout, dpos, (long)B, (long)S, d);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ----------------------------------------------------------------------------- cross entropy
// one 256-thread block per row; online (max, sum-exp), sum of logits, target logit, argmax.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const unsigned short* logits, long ld, const long* targets, long V,
                                                      float eps, float* row_loss, float* row_lse, long* argmax) {
  __shared__ float sm[4], ss[4], sx[4], sbv[4];
  __shared__ int sbi[4];
  const long row = blockIdx.x;
  const unsigned short* x = logits + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float m = -3.0e38f, s = 0.f, sumx = 0.f, bv = -3.0e38f;
  int bi = 0x7fffffff;
  const long nv = V >> 3;
  for (long i = tid; i < nv; i += 256) {
    const u32x4 r = *(const u32x4*)(x + i * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bf2f((unsigned short)(r[e] & 0xffff));
      v[2 * e + 1] = bf2f((unsigned short)(r[e] >> 16));
    }
    float cm = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, v[e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (v[e] > bv) { bv = v[e]; bi = (int)(i * 8 + e); }
      sumx += v[e];
    }
    if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(v[e] - m);
  }
  for (long c = (nv << 3) + tid; c < V; c += 256) {
    const float v = bf2f(x[c]);
    if (v > bv) { bv = v; bi = (int)c; }
    sumx += v;
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  // wave reduce
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
    const float nm = fmaxf(m, om);
    s = s * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
    sumx += __shfl_xor(sumx, o, 64);
    const float obv = __shfl_xor(bv, o, 64);
    const int obi = __shfl_xor(bi, o, 64);
    if (obv > bv || (obv == bv && obi < bi)) { bv = obv; bi = obi; }
  }
  if (lane == 0) { sm[wv] = m; ss[wv] = s; sx[wv] = sumx; sbv[wv] = bv; sbi[wv] = bi; }
  __syncthreads();
  if (tid == 0) {
    float M = sm[0], S = ss[0], X = sx[0], BV = sbv[0];
    int BI = sbi[0];
    for (int w = 1; w < 4; ++w) {
      const float nm = fmaxf(M, sm[w]);
      S = S * __expf(M - nm) + ss[w] * __expf(sm[w] - nm);
      M = nm;
      X += sx[w];
      if (sbv[w] > BV || (sbv[w] == BV && sbi[w] < BI)) { BV = sbv[w]; BI = sbi[w]; }
    }
    const float lse = M + __logf(S);
    const long t = targets[row];
    float loss = 0.f;
    if (t >= 0 && t < V) {
      const float xt = bf2f(x[t]);
      loss = (1.f - eps) * (lse - xt) + eps * (lse - X / (float)V);
    }
    row_loss[row] = loss;
    row_lse[row] = lse;
    if (argmax) argmax[row] = BI;
  }
}
// deterministic reduction of the row losses (single block)
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss, const long* targets, long rows, long V,
                                                         float* stats) {
  __shared__ float sl[256], sc[256];
  float l = 0.f, c = 0.f;
  for (long r = threadIdx.x; r < rows; r += 256) {
    const long t = targets[r];
    if (t >= 0 && t < V) { l += row_loss[r]; c += 1.f; }
  }
  sl[threadIdx.x] = l;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { stats[0] = sl[0]; stats[1] = sc[0]; }
}
extern "C" int wft_ce_fwd(const wft_bf16* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V,
                          float label_smoothing, float* row_loss, float* row_lse, float* stats, int64_t* argmax,
                          void* stream) {
  WFT_CHECK_ARG(logits && targets && row_loss && row_lse && stats, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && V >= 1 && ld >= V && ld % 8 == 0, "bad shape (ld must be a multiple of 8, >= V)");
  WFT_CHECK_ARG((((uintptr_t)logits) & 15) == 0, "16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, (long)ld, (const long*)targets, (long)V,
                     label_smoothing, row_loss, row_lse, (long*)argmax);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, (const long*)targets, (long)rows, (long)V, stats);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const unsigned short* logits, long ld, const long* targets, long V,
                                                      float eps, const float* row_lse, const float* stats,
                                                      const float* gscale, unsigned short* dlogits) {
  const long row = blockIdx.x;
  const unsigned short* x = logits + row * ld;
  unsigned short* dx = dlogits + row * ld;
  const long t = targets[row];
  const bool valid = t >= 0 && t < V;
  const float coef = valid ? gscale[0] / fmaxf(stats[1], 1.f) : 0.f;
  const float lse = row_lse[row];
  const float sm = eps / (float)V;
  const long nv = ld >> 3;
  for (long i = threadIdx.x; i < nv; i += 256) {
    const u32x4 r = *(const u32x4*)(x + i * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bf2f((unsigned short)(r[e] & 0xffff));
      v[2 * e + 1] = bf2f((unsigned short)(r[e] >> 16));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long c = i * 8 + e;
      float g = 0.f;
      if (c < V) {
        g = __expf(v[e] - lse) - sm;
        if (c == t) g -= (1.f - eps);
        g *= coef;
      }
      v[e] = g;
    }
    u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *(u32x4*)(dx + i * 8) = o;
  }
}
extern "C" int wft_ce_bwd(const wft_bf16* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V,
                          float label_smoothing, const float* row_lse, const float* stats, const float* gscale,
                          wft_bf16* dlogits, void* stream) {
  WFT_CHECK_ARG(logits && targets && row_lse && stats && gscale && dlogits, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && V >= 1 && ld >= V && ld % 8 == 0, "bad shape (ld must be a multiple of 8, >= V)");
  WFT_CHECK_ARG((((uintptr_t)logits) & 15) == 0 && (((uintptr_t)dlogits) & 15) == 0, "16-byte alignment");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld,
                     (const long*)targets, (long)V, label_smoothing, row_lse, stats, gscale, dlogits);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ----------------------------------------------------------------------------- eval token statistics
// One pass over the V logits of every token (eval/metrics.py:106-137 computes log_softmax, softmax, CE,
// entropy and max-prob as five separate [S, V] passes): per row
//   out[row] = { lse, max logit, sum_c p_c * x_c, x_target (0 if ignored) },  argmax[row] (lowest index on ties)
// from which nll = lse - x_t, log p(pred) = max - lse, confidence = exp(max - lse), entropy = lse - E_p[x].
__global__ __launch_bounds__(256) void token_stats_kernel(const unsigned short* logits, long ld, const long* targets, long V,
                                                           float* out4, long* argmax) {
  __shared__ float sm[4], ss[4], sw[4], sbv[4];
  __shared__ int sbi[4];
  const long row = blockIdx.x;
  const unsigned short* x = logits + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float m = -3.0e38f, s = 0.f, w = 0.f, bv = -3.0e38f;  // w = sum exp(x - m) * x
  int bi = 0x7fffffff;
  auto push = [&](float v, int idx) {
    if (v > bv) { bv = v; bi = idx; }
    if (v > m) { const float f = __expf(m - v); s *= f; w *= f; m = v; }
    const float e = __expf(v - m);
    s += e;
    w += e * v;
  };
  const long nv = V >> 3;
  for (long i = tid; i < nv; i += 256) {
    const u32x4 r = *(const u32x4*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      push(bf2f((unsigned short)(r[e] & 0xffff)), (int)(i * 8 + 2 * e));
      push(bf2f((unsigned short)(r[e] >> 16)), (int)(i * 8 + 2 * e + 1));
    }
  }
  for (long c = (nv << 3) + tid; c < V; c += 256) push(bf2f(x[c]), (int)c);
  auto merge = [&](float om, float os, float ow, float obv, int obi) {
    const float nm = fmaxf(m, om);
    const float f0 = __expf(m - nm), f1 = __expf(om - nm);
    s = s * f0 + os * f1;
    w = w * f0 + ow * f1;
    m = nm;
    if (obv > bv || (obv == bv && obi < bi)) { bv = obv; bi = obi; }
  };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    merge(__shfl_xor(m, o, 64), __shfl_xor(s, o, 64), __shfl_xor(w, o, 64), __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
  if (lane == 0) { sm[wv] = m; ss[wv] = s; sw[wv] = w; sbv[wv] = bv; sbi[wv] = bi; }
  __syncthreads();
  if (tid == 0) {
    m = sm[0]; s = ss[0]; w = sw[0]; bv = sbv[0]; bi = sbi[0];
    for (int k = 1; k < 4; ++k) merge(sm[k], ss[k], sw[k], sbv[k], sbi[k]);
    const long t = targets ? targets[row] : -100;
    out4[row * 4 + 0] = m + __logf(s);
    out4[row * 4 + 1] = bv;
    out4[row * 4 + 2] = w / s;
    out4[row * 4 + 3] = (t >= 0 && t < V) ? bf2f(x[t]) : 0.f;
    argmax[row] = bi;
  }
}
extern "C" int wft_token_stats(const wft_bf16* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V,
                               float* out4, int64_t* argmax, void* stream) {
  WFT_CHECK_ARG(logits && out4 && argmax, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && V >= 1 && ld >= V && ld % 8 == 0, "bad shape (ld must be a multiple of 8, >= V)");
  WFT_CHECK_ARG((((uintptr_t)logits) & 15) == 0, "16-byte alignment");
  hipLaunchKernelGGL(token_stats_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld,
                     (const long*)targets, (long)V, out4, (long*)argmax);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ----------------------------------------------------------------------------- AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, unsigned short* pb,
                                                     long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                     float bc2, const float* gscale) {
  const float gs = gscale ? gscale[0] : 1.f;
  const float step = lr / bc1;
  const float rbc2 = rsqrtf(bc2);
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    f32x4 pp = *(f32x4*)(p + i * 4);
    const f32x4 gg = *(const f32x4*)(g + i * 4) * gs;
    f32x4 mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= (1.f - lr * wd);
      mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
      vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
      const float denom = sqrtf(vv[e]) * rbc2 + eps;
      pp[e] -= step * mm[e] / denom;
    }
    *(f32x4*)(p + i * 4) = pp;
    *(f32x4*)(m + i * 4) = mm;
    *(f32x4*)(v + i * 4) = vv;
    if (pb) {
      u32x2 o = {pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
      *(u32x2*)(pb + i * 4) = o;
    }
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 2) + threadIdx.x;
    if (t < n) {
      float pp = p[t] * (1.f - lr * wd);
      const float gg = g[t] * gs;
      const float mm = b1 * m[t] + (1.f - b1) * gg;
      const float vv = b2 * v[t] + (1.f - b2) * gg * gg;
      pp -= step * mm / (sqrtf(vv) * rbc2 + eps);
      p[t] = pp; m[t] = mm; v[t] = vv;
      if (pb) pb[t] = f2bf(pp);
    }
  }
}
extern "C" int wft_adamw_step(float* p, const float* g, float* m, float* v, wft_bf16* p_bf16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                              const float* gscale, void* stream) {
  WFT_CHECK_ARG(p && g && m && v && n >= 1, "bad args");
  WFT_CHECK_ARG((((uintptr_t)p) & 15) == 0 && (((uintptr_t)g) & 15) == 0 && (((uintptr_t)m) & 15) == 0 &&
                    (((uintptr_t)v) & 15) == 0 && (!p_bf16 || (((uintptr_t)p_bf16) & 7) == 0),
                "16-byte alignment");
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, p_bf16, (long)n,
                     lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, gscale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const f32x4 x = *(const f32x4*)(g + i * 4);
    s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  if (blockIdx.x == 0) {
    const long t = (nv << 2) + threadIdx.x;
    if (t < n) s += g[t] * g[t];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
extern "C" int wft_sumsq_f32(const float* g, int64_t n, float* out, void* stream) {
  WFT_CHECK_ARG(g && out && n >= 1, "bad args");
  WFT_CHECK_ARG((((uintptr_t)g) & 15) == 0, "16-byte alignment");
  int grid = ew_grid(n / 4 + 1);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, (long)n, out);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// n small f32 copies in one launch (tab: int64 [n][3] = source address, destination address, element count): the stacked bias
// vectors of the fused q/k/v groups after an optimizer step.  160 torch copy_ calls per step did this before, each a blit
// launch of its own with ~40 us between two of them (profiles/r03_headline_gap_analysis.log).
__global__ __launch_bounds__(256) void mt_copy_f32_kernel(const long* tab) {
  const long* row = tab + 3 * (long)blockIdx.x;
  const float* src = (const float*)row[0];
  float* dst = (float*)row[1];
  const long n = row[2] & 0xffffffffL;
  const int fsb = (int)(row[2] >> 32);  // bits 32..63 of the count field: a scale as f32 bits (0 = plain copy): the q slice of a fused bias
  if (fsb) {
    const float fs = __int_as_float(fsb);
    for (long i = threadIdx.x; i < n; i += 256) dst[i] = src[i] * fs;
  } else {
    for (long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
  }
}
extern "C" int wft_mt_copy_f32(const void* tab, int n, void* stream) {
  WFT_CHECK_ARG(tab && n >= 1, "bad args");
  hipLaunchKernelGGL(mt_copy_f32_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const long*)tab);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
