// norm.hip — LayerNorm forward / backward (HBM-bound; one wave per row, 16-byte loads,
// wave-shuffle reductions, fp32 statistics as whisper.model.LayerNorm does).
// Algorithmic bytes: fwd 2*rows*cols*2 B (read x, write y); bwd 3 reads + 1 write.
#include "common.h"

#define LN_MAXC 4  // chunks of 8 bf16 per lane -> cols <= 64*8*4 = 2048

__device__ __forceinline__ void load8(const unsigned short* p, float* v) {
  const u32x4 r = *(const u32x4*)p;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = bf2f((unsigned short)(r[e] & 0xffff));
    v[2 * e + 1] = bf2f((unsigned short)(r[e] >> 16));
  }
}
__device__ __forceinline__ void store8(unsigned short* p, const float* v) {
  u32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = pack2bf(v[2 * e], v[2 * e + 1]);
  *(u32x4*)p = r;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const unsigned short* x, const float* gamma,
                                                      const float* beta, unsigned short* y, float* mean,
                                                      float* rstd, long rows, int cols, float eps, int rpb,
                                                      int t0, int t1, int c0, int c1) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  const int nch = cols >> 3;
  for (long row = wave_id; row < rows; row += nwaves) {
    const unsigned short* xr = x + row * cols;
    float v[LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        load8(xr + ch * 8, v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
      }
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[c][e] - mu;
          q += d * d;
        }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    bool trow = false;
    if (rpb > 0) {
      const int t = (int)(row % rpb);
      trow = (t >= t0 && t < t1);
    }
    unsigned short* yr = y + row * cols;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float o[8];
        const f32x4 g0 = *(const f32x4*)(gamma + ch * 8), g1 = *(const f32x4*)(gamma + ch * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(beta + ch * 8), b1 = *(const f32x4*)(beta + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gg = e < 4 ? g0[e] : g1[e - 4];
          const float bb = e < 4 ? b0[e] : b1[e - 4];
          o[e] = (v[c][e] - mu) * rs * gg + bb;
          if (rpb > 0) {
            const int col = ch * 8 + e;
            if (trow || (col >= c0 && col < c1)) o[e] = 0.f;
          }
        }
        store8(yr + ch * 8, o);
      }
    }
  }
}

// backward: each wave walks rows (grid-stride), keeps per-lane dgamma/dbeta partials for its
// columns, then the 4 waves of a block are summed through LDS and written to
// partial[block][2][cols]; ln_bwd_reduce sums the blocks (deterministic, no atomics).
// NC = ceil(cols / 512): per-lane column chunks kept in registers (3 for d = 1280: 2 waves/SIMD even with the dx column sums)
template <bool DXSUM, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const unsigned short* dy, const unsigned short* x,
                                                      const float* gamma, const float* mean,
                                                      const float* rstd, const unsigned short* dres,
                                                      unsigned short* dx, float* partial, long rows, int cols,
                                                      int rpb, int t0, int t1, int c0, int c1) {
  __shared__ float red[4][DXSUM ? 3 : 2][512];  // one 512-column slot (64 lanes x 8), reused per chunk pass
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave_id = (long)blockIdx.x * 4 + wv;
  const long nwaves = (long)gridDim.x * 4;
  const int nch = cols >> 3;
  float dg[NC][8], db[NC][8], gm[NC][8];
  float dsum[DXSUM ? NC : 1][8];  // column sums of the bf16 dx this wave writes (bias grad of the producing Linear)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (DXSUM) dsum[c][e] = 0.f;
      dg[c][e] = 0.f;
      db[c][e] = 0.f;
      gm[c][e] = (ch < nch) ? gamma[ch * 8 + e] : 0.f;
    }
  }
  for (long row = wave_id; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    bool trow = false;
    if (rpb > 0) {
      const int t = (int)(row % rpb);
      trow = (t >= t0 && t < t1);
    }
    float g[NC][8], xh[NC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float d[8], xv[8];
        load8(dy + row * cols + ch * 8, d);
        load8(x + row * cols + ch * 8, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (rpb > 0) {
            const int col = ch * 8 + e;
            if (trow || (col >= c0 && col < c1)) d[e] = 0.f;
          }
          xh[c][e] = (xv[e] - mu) * rs;
          g[c][e] = d[e] * gm[c][e];
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
          dg[c][e] += d[e] * xh[c][e];
          db[c][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) / cols;
    s2 = wave_sum(s2) / cols;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (g[c][e] - s1 - xh[c][e] * s2);
        if (dres) {
          float rv[8];
          load8(dres + row * cols + ch * 8, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += rv[e];
        }
        if (DXSUM) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dsum[c][e] += bf2f(f2bf(o[e]));  // what wft_colsum_bf16 would read back
        }
        store8(dx + row * cols + ch * 8, o);
      }
    }
  }
  // block reduction of dgamma/dbeta partials, one chunk-slot at a time through LDS
  float* pg = partial + (long)blockIdx.x * (DXSUM ? 3 : 2) * cols;
  float* pb = pg + cols;
  float* ps = pb + cols;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wv][0][lane * 8 + e] = dg[c][e];
      red[wv][1][lane * 8 + e] = db[c][e];
      if (DXSUM) red[wv][2][lane * 8 + e] = dsum[c][e];
    }
    __syncthreads();
    // 512 columns of this slot, 256 threads -> 2 each
    for (int k = threadIdx.x; k < 512; k += 256) {
      const int col = c * 512 + k;
      if (col < cols) {
        pg[col] = red[0][0][k] + red[1][0][k] + red[2][0][k] + red[3][0][k];
        pb[col] = red[0][1][k] + red[1][1][k] + red[2][1][k] + red[3][1][k];
        if (DXSUM) ps[col] = red[0][2][k] + red[1][2][k] + red[2][2][k] + red[3][2][k];
      }
    }
  }
}

// sums the per-block partials: 64 columns per workgroup (256-byte row segments), 4 waves stride the partial rows of
// this workgroup's row chunk (gridDim.y chunks).  Run twice: [nblocks] -> [LN_RED_CHUNKS] -> final (+= dgamma, dbeta; = dx sums).
#define LN_RED_CHUNKS 16
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* partial, int nblocks, int cols, int nset,
                                                             float* mid, float* dgamma, float* dbeta, float* dxsum) {
  __shared__ float red[3][4][64];
  const int cx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  float acc[3] = {0.f, 0.f, 0.f};
  if (col < cols)
    for (int b = b0 + wv; b < b1; b += 4) {
      const float* pp = partial + (long)b * nset * cols + col;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < nset) acc[k] += pp[(long)k * cols];
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][wv][cx] = acc[k];
  __syncthreads();
  if (wv == 0 && col < cols) {
    float tot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) tot[k] = red[k][0][cx] + red[k][1][cx] + red[k][2][cx] + red[k][3][cx];
    if (mid) {
      float* mp = mid + (long)blockIdx.y * nset * cols + col;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < nset) mp[(long)k * cols] = tot[k];
    } else {
      dgamma[col] += tot[0];
      dbeta[col] += tot[1];
      if (dxsum) dxsum[col] = tot[2];
    }
  }
}

static int ln_grid(long rows) {
  long g = (rows + 3) / 4;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int wft_layernorm_fwd(const wft_bf16* x, const float* gamma, const float* beta, wft_bf16* y,
                                 float* mean, float* rstd, int64_t rows, int cols, float eps,
                                 int rows_per_batch, int t0, int t1, int c0, int c1, void* stream) {
  WFT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && cols <= 2048, "cols must be a multiple of 8, <= 2048");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, (long)rows, cols, eps, rows_per_batch, t0, t1, c0, c1);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

extern "C" int64_t wft_layernorm_bwd_workspace(int64_t rows, int cols) {
  return ((int64_t)ln_grid(rows) + LN_RED_CHUNKS) * 3 * cols * sizeof(float);
}

extern "C" int wft_layernorm_bwd(const wft_bf16* dy, const wft_bf16* x, const float* gamma, const float* mean,
                                 const float* rstd, const wft_bf16* dres, wft_bf16* dx, float* dgamma,
                                 float* dbeta, float* dx_colsum, void* partial, int64_t rows, int cols,
                                 int rows_per_batch, int t0, int t1, int c0, int c1, void* stream) {
  WFT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && partial, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && cols <= 2048, "cols must be a multiple of 8, <= 2048");
  const int grid = ln_grid(rows);
#define LN_BWD_LAUNCH(DX, NCV)                                                                                     \
  hipLaunchKernelGGL((ln_bwd_kernel<DX, NCV>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, \
                     dres, dx, (float*)partial, (long)rows, cols, rows_per_batch, t0, t1, c0, c1)
  const int nc = (cols + 511) / 512;
  if (dx_colsum) {
    if (nc == 1) LN_BWD_LAUNCH(true, 1); else if (nc == 2) LN_BWD_LAUNCH(true, 2); else if (nc == 3) LN_BWD_LAUNCH(true, 3); else LN_BWD_LAUNCH(true, 4);
  } else {
    if (nc == 1) LN_BWD_LAUNCH(false, 1); else if (nc == 2) LN_BWD_LAUNCH(false, 2); else if (nc == 3) LN_BWD_LAUNCH(false, 3); else LN_BWD_LAUNCH(false, 4);
  }
#undef LN_BWD_LAUNCH
  const int nset = dx_colsum ? 3 : 2;
  float* mid = (float*)partial + (long)grid * nset * cols;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64, LN_RED_CHUNKS), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partial, grid, cols, nset, mid, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, (const float*)mid,
                     LN_RED_CHUNKS, cols, nset, (float*)nullptr, dgamma, dbeta, dx_colsum);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
