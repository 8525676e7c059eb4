// norm.hip — LayerNorm forward / backward (HBM-bound; one wave per row, next row prefetched,
// wave-shuffle reductions, fp32 statistics as whisper.model.LayerNorm does).
// Algorithmic bytes: fwd 2*rows*cols*2 B (read x, write y); bwd 3 reads + 1 write.
#include "common.h"

// Column mapping: 4-column pieces (8-byte loads), lane l owns pieces l, l+64, ... (NQ = ceil(cols / 256) of them: exactly
// 5 for d = 1280, 2 for 512; a wave instruction still covers 512 contiguous bytes of the row).
__device__ __forceinline__ void cvt4(const u32x2 r, float* v) {
  v[0] = bf2f((unsigned short)(r[0] & 0xffff));
  v[1] = bf2f((unsigned short)(r[0] >> 16));
  v[2] = bf2f((unsigned short)(r[1] & 0xffff));
  v[3] = bf2f((unsigned short)(r[1] >> 16));
}
__device__ __forceinline__ void store4(unsigned short* p, const float* v) {
  u32x2 r;
  r[0] = pack2bf(v[0], v[1]);
  r[1] = pack2bf(v[2], v[3]);
  *(u32x2*)p = r;
}

// forward: one wave per row; gamma/beta stay in registers and the next row's loads are in flight while the current row
// is reduced and written (software prefetch), 4 waves per SIMD at NQ = 5
template <int NQ, int VAR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const unsigned short* x, const float* gamma,
                                                      const float* beta, unsigned short* y, float* mean,
                                                      float* rstd, long rows, int cols, float eps, int rpb,
                                                      int t0, int t1, int c0, int c1) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  const int nch = cols >> 2;
  const float inv = 1.f / cols;
  f32x4 gg[NQ], bb[NQ];
  if (VAR >= 1) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      gg[c] = bb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ch < nch) {
        gg[c] = *(const f32x4*)(gamma + ch * 4);
        bb[c] = *(const f32x4*)(beta + ch * 4);
      }
    }
  }
  u32x2 nraw[NQ];
  auto fetch = [&](long row) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      nraw[c] = u32x2{0u, 0u};
      if (ch < nch) nraw[c] = *(const u32x2*)(x + row * cols + ch * 4);
    }
  };
  if (wave_id < rows) fetch(wave_id);
  for (long row = wave_id; row < rows; row += nwaves) {
    float v[NQ][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      cvt4(nraw[c], v[c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[c][e];
    }
    if (VAR >= 1 && row + nwaves < rows) fetch(row + nwaves);
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      if (lane + c * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[c][e] - mu;
          q += d * d;
        }
      }
    }
    const float rs = rsqrtf(wave_sum(q) * inv + eps);
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
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float o[4];
        if (VAR == 0) {
          gg[c] = *(const f32x4*)(gamma + ch * 4);
          bb[c] = *(const f32x4*)(beta + ch * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (v[c][e] - mu) * rs * gg[c][e] + bb[c][e];
          if (rpb > 0) {
            const int col = ch * 4 + e;
            if (trow || (col >= c0 && col < c1)) o[e] = 0.f;
          }
        }
        if (VAR == 2) {
          u32x2 r;
          r[0] = pack2bf(o[0], o[1]);
          r[1] = pack2bf(o[2], o[3]);
          __builtin_nontemporal_store(r, (u32x2*)(yr + ch * 4));
        } else {
          store4(yr + ch * 4, o);
        }
      }
    }
    if (VAR == 0 && row + nwaves < rows) fetch(row + nwaves);
  }
}

// backward: each wave walks rows (grid-stride), keeps per-lane dgamma/dbeta partials for its columns, then the 4 waves
// of a block are summed through LDS and written to partial[block][2 or 3][cols]; ln_bwd_reduce sums the blocks
// (deterministic, no atomics).  The register file allows two waves per SIMD at d = 1280, so every wave keeps the NEXT
// row's dy / x / dres loads in flight while it reduces and writes the current one (software prefetch).
template <bool DXSUM, int NQ, int VAR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const unsigned short* dy, const unsigned short* x,
                                                      const float* gamma, const float* mean,
                                                      const float* rstd, const unsigned short* dres,
                                                      unsigned short* dx, float* partial, long rows, int cols,
                                                      int rpb, int t0, int t1, int c0, int c1) {
  __shared__ float red[4][DXSUM ? 3 : 2][256];  // one 256-column slot (64 lanes x 4), reused per piece pass
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave_id = (long)blockIdx.x * 4 + wv;
  const long nwaves = (long)gridDim.x * 4;
  const int nch = cols >> 2;
  const float inv = 1.f / cols;
  float dg[NQ][4], db[NQ][4], gm[NQ][4];
  float dsum[DXSUM ? NQ : 1][4];  // column sums of the bf16 dx this wave writes (bias grad of the producing Linear)
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (DXSUM) dsum[c][e] = 0.f;
      dg[c][e] = 0.f;
      db[c][e] = 0.f;
      gm[c][e] = (ch < nch) ? gamma[ch * 4 + e] : 0.f;
    }
  }
  u32x2 ndy[NQ], nx[NQ], nr[NQ];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](long row) {
    nmu = mean[row];
    nrs = rstd[row];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      ndy[c] = nx[c] = nr[c] = u32x2{0u, 0u};
      if (ch < nch) {
        if (VAR == 2) {
          ndy[c] = __builtin_nontemporal_load((const u32x2*)(dy + row * cols + ch * 4));
          nx[c] = __builtin_nontemporal_load((const u32x2*)(x + row * cols + ch * 4));
          if (dres) nr[c] = __builtin_nontemporal_load((const u32x2*)(dres + row * cols + ch * 4));
        } else {
          ndy[c] = *(const u32x2*)(dy + row * cols + ch * 4);
          nx[c] = *(const u32x2*)(x + row * cols + ch * 4);
          if (dres) nr[c] = *(const u32x2*)(dres + row * cols + ch * 4);
        }
      }
    }
  };
  if (wave_id < rows) fetch(wave_id);
  for (long row = wave_id; row < rows; row += nwaves) {
    const float mu = nmu, rs = nrs;
    u32x2 rdy[NQ], rx[NQ], rr[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      rdy[c] = ndy[c];
      rx[c] = nx[c];
      rr[c] = nr[c];
    }
    if (row + nwaves < rows) fetch(row + nwaves);
    bool trow = false;
    if (rpb > 0) {
      const int t = (int)(row % rpb);
      trow = (t >= t0 && t < t1);
    }
    float g[NQ][4], xh[NQ][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      float d[4], xv[4];
      cvt4(rdy[c], d);
      cvt4(rx[c], xv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (rpb > 0) {
          const int col = ch * 4 + e;
          if (trow || (col >= c0 && col < c1)) d[e] = 0.f;
        }
        // explicit fmas: both instantiations (with / without dx column sums) must round identically
        xh[c][e] = (ch < nch) ? (xv[e] - mu) * rs : 0.f;
        g[c][e] = d[e] * gm[c][e];
        s1 += g[c][e];
        s2 = __builtin_fmaf(g[c][e], xh[c][e], s2);
        dg[c][e] = __builtin_fmaf(d[e], xh[c][e], dg[c][e]);
        db[c][e] += d[e];
      }
    }
    s1 = wave_sum(s1) * inv;
    s2 = wave_sum(s2) * inv;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = lane + c * 64;
      if (ch < nch) {
        float o[4], rv[4];
        cvt4(rr[c], rv);  // zeros without a residual gradient
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(rs, __builtin_fmaf(-xh[c][e], s2, g[c][e] - s1), rv[e]);
        if (DXSUM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dsum[c][e] += bf2f(f2bf(o[e]));  // what wft_colsum_bf16 would read back
        }
        if (VAR >= 1) {
          u32x2 r;
          r[0] = pack2bf(o[0], o[1]);
          r[1] = pack2bf(o[2], o[3]);
          __builtin_nontemporal_store(r, (u32x2*)(dx + row * cols + ch * 4));
        } else {
          store4(dx + row * cols + ch * 4, o);
        }
      }
    }
  }
  // block reduction of dgamma/dbeta partials, one piece-slot at a time through LDS
  if (partial == nullptr) return;  // frozen gamma / beta and no column sums wanted (a LoRA run): dx was everything
  float* pg = partial + (long)blockIdx.x * (DXSUM ? 3 : 2) * cols;
  float* pb = pg + cols;
  float* ps = pb + cols;
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wv][0][lane * 4 + e] = dg[c][e];
      red[wv][1][lane * 4 + e] = db[c][e];
      if (DXSUM) red[wv][2][lane * 4 + e] = dsum[c][e];
    }
    __syncthreads();
    const int k = threadIdx.x;  // 256 columns of this slot, one per thread
    const int col = c * 256 + k;
    if (col < cols) {
      pg[col] = red[0][0][k] + red[1][0][k] + red[2][0][k] + red[3][0][k];
      pb[col] = red[0][1][k] + red[1][1][k] + red[2][1][k] + red[3][1][k];
      if (DXSUM) ps[col] = red[0][2][k] + red[1][2][k] + red[2][2][k] + red[3][2][k];
    }
  }
}

// sums the per-block partials: 64 columns per workgroup (256-byte row segments), 4 waves stride the partial rows of
// this workgroup's row chunk (gridDim.y chunks).  Run twice: [nblocks] -> [LN_RED_CHUNKS] -> final (dgamma, dbeta, dx sums written).
#define LN_RED_CHUNKS 16
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* partial, int nblocks, int cols, int nset,
                                                             float* mid, float* dgamma, float* dbeta, float* dxsum) {
  __shared__ float red[3][4][64];
  const int cx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  float acc[3] = {0.f, 0.f, 0.f};
  if (col < cols) {
    int b = b0 + wv;
    // eight partial rows' loads in flight per wave (the single-level call of small problems walks up to 128 rows per wave: one
    // exposed load latency per row otherwise); same order of additions as the plain loop
    for (; b + 28 < b1; b += 32) {
      float v[8][3];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* pp = partial + (long)(b + 4 * u) * nset * cols + col;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[u][k] = k < nset ? pp[(long)k * cols] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nset) acc[k] += v[u][k];
    }
    for (; b < b1; b += 4) {
      const float* pp = partial + (long)b * nset * cols + col;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < nset) acc[k] += pp[(long)k * cols];
    }
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
      if (dgamma) {
        dgamma[col] = tot[0];
        dbeta[col] = tot[1];
      }
      if (dxsum) dxsum[col] = tot[2];
    }
  }
}

static int ln_grid(long rows) {  // backward: 2 waves per SIMD (register-bound), and one partial row per workgroup
  long g = (rows + 3) / 4;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}
// forward: as many workgroups as are resident at once (occupancy of the instantiation, queried once), grid-stride rows
template <int NQ, int VAR>
static int ln_fwd_grid(long rows) {
  static int resident = 0;
  if (!resident) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_fwd_kernel<NQ, VAR>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    resident = per_cu * 256;
  }
  long g = (rows + 3) / 4;
  if (g > resident) g = resident;
  if (g < 1) g = 1;
  return (int)g;
}

#define LN_NT_BYTES (64L << 20)
#define LN_NQ_SWITCH(nq, M) \
  switch (nq) { case 1: M(1); break; case 2: M(2); break; case 3: M(3); break; case 4: M(4); break; \
                case 5: M(5); break; case 6: M(6); break; case 7: M(7); break; default: M(8); break; }

extern "C" int wft_layernorm_fwd(const wft_bf16* x, const float* gamma, const float* beta, wft_bf16* y,
                                 float* mean, float* rstd, int64_t rows, int cols, float eps,
                                 int rows_per_batch, int t0, int t1, int c0, int c1, void* stream) {
  WFT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "null pointer");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && cols <= 2048, "cols must be a multiple of 8, <= 2048");
  // VAR 1: gamma/beta in registers + next-row prefetch; VAR 2: the same with non-temporal y stores, used when the
  // tensor is larger than the caches could keep for the consumer anyway (5.4 -> 6.5 TB/s at 68 x 1500 x 1280);
  // VAR 0 (no prefetch, 3.4 TB/s) only through WFT_LN_FWD_VAR for A/B runs
  static int forced = -2;
  if (forced == -2) { const char* e = wft_dev_getenv("WFT_LN_FWD_VAR"); forced = e ? atoi(e) : -1; }
  const int var = forced >= 0 ? forced : ((long)rows * cols * 2 >= LN_NT_BYTES ? 2 : 1);
#define LN_FWD_LAUNCH_V(NQV, V)                                                                                     \
  hipLaunchKernelGGL((ln_fwd_kernel<NQV, V>), dim3(ln_fwd_grid<NQV, V>(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, \
                     beta, y, mean, rstd, (long)rows, cols, eps, rows_per_batch, t0, t1, c0, c1)
#define LN_FWD_LAUNCH(NQV) \
  if (var == 0) LN_FWD_LAUNCH_V(NQV, 0); else if (var == 2) LN_FWD_LAUNCH_V(NQV, 2); else LN_FWD_LAUNCH_V(NQV, 1)
  LN_NQ_SWITCH((cols + 255) / 256, LN_FWD_LAUNCH)
#undef LN_FWD_LAUNCH
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
  WFT_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "null pointer");
  WFT_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr), "dgamma and dbeta go together (both or neither)");
  const bool want_params = dgamma != nullptr;
  WFT_CHECK_ARG(partial || (!want_params && !dx_colsum), "the partial-sum workspace is required unless dgamma, dbeta and dx_colsum are all NULL");
  WFT_CHECK_ARG(rows >= 1 && cols >= 8 && cols % 8 == 0 && cols <= 2048, "cols must be a multiple of 8, <= 2048");
  const int grid = ln_grid(rows);
  if (!want_params && !dx_colsum) partial = nullptr;  // nothing to reduce: the kernel stops after dx, no reduce launches
  // VAR 2 (non-temporal dy / x / dres loads and dx stores) for tensors the caches cannot keep; WFT_LN_BWD_VAR forces one
  static int forced = -2;
  if (forced == -2) { const char* e = wft_dev_getenv("WFT_LN_BWD_VAR"); forced = e ? atoi(e) : -1; }
  const int var = forced >= 0 ? forced : ((long)rows * cols * 2 >= LN_NT_BYTES ? 2 : 0);
#define LN_BWD_LAUNCH_V(DX, NCV, V)                                                                                 \
  hipLaunchKernelGGL((ln_bwd_kernel<DX, NCV, V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, \
                     dres, dx, (float*)partial, (long)rows, cols, rows_per_batch, t0, t1, c0, c1)
#define LN_BWD_LAUNCH(DX, NCV) \
  if (var == 0) LN_BWD_LAUNCH_V(DX, NCV, 0); else if (var == 2) LN_BWD_LAUNCH_V(DX, NCV, 2); else LN_BWD_LAUNCH_V(DX, NCV, 1)
  const int nq = (cols + 255) / 256;
#define LN_BWD_T(NQV) LN_BWD_LAUNCH(true, NQV)
#define LN_BWD_F(NQV) LN_BWD_LAUNCH(false, NQV)
  if (dx_colsum) { LN_NQ_SWITCH(nq, LN_BWD_T) } else { LN_NQ_SWITCH(nq, LN_BWD_F) }
#undef LN_BWD_T
#undef LN_BWD_F
#undef LN_BWD_LAUNCH
  if (!partial) {
    WFT_CHECK_LAUNCH();
    return WFT_OK;
  }
  const int nset = dx_colsum ? 3 : 2;
  if (grid <= 256) {
    // small calls (a decoder block at B*S = 1 024 rows: 38 of the 64 LayerNorm backward calls of a whisper-base step) are launch-bound:
    // one reduce launch over the <= 256 partial rows instead of two levels (16.2 -> 13.9 us per call; from 512 partial rows on the
    // two-level form wins: 21.6 against 26.1 us at 12 000 x 512; tools/dev/ln_time.py)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, grid,
                       cols, nset, (float*)nullptr, dgamma, dbeta, dx_colsum);
    WFT_CHECK_LAUNCH();
    return WFT_OK;
  }
  float* mid = (float*)partial + (long)grid * nset * cols;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64, LN_RED_CHUNKS), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partial, grid, cols, nset, mid, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, (const float*)mid,
                     LN_RED_CHUNKS, cols, nset, (float*)nullptr, dgamma, dbeta, dx_colsum);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
