// f32.hip — the fp32 compute mode of the path (training.mixed_precision_training: False).
//
// The reference runs true fp32 when AMP is off (model/model_utils.py:37-48,64: autocast disabled, every Linear / conv /
// attention / LayerNorm / cross-entropy in fp32) and BASELINE.json's north star asks for loss / metric parity within 1e-3
// relative in that mode.  These kernels are that mode: fp32 tensors in HBM, fp32 MFMA (v_mfma_f32_32x32x2_f32, full fp32
// multiplies and accumulation — not the bf16 or xf32 pipes), fp32 everywhere else.  It is the PARITY mode (whisper-tiny /
// base configurations, configs[0]); nothing here is tuned beyond coalesced access and a fixed summation order (bitwise
// reproducible): the throughput path is the bf16 one (gemm.hip, attn.hip, norm.hip).
//
//   wft_gemm_f32        C = alpha * op(A) op(B) (+ beta * C) (+ bias[n]), batched, arbitrary element strides for A and B
//                       (NT / TN / NN, and the conv stem's overlapping-window rows without an im2col copy)
//   wft_softmax_f32     in-place row softmax of scale * s (+ causal mask)     / its backward
//   wft_layernorm_*_f32 LayerNorm with the deep-SpecAugment mask              / backward (dx, dgamma, dbeta)
//   wft_gelu_*_f32, wft_axpby_f32, wft_colsum_f32, wft_embed_*_f32, wft_ce_*_f32
#include "common.h"

// ------------------------------------------------------------------------------------------------ GEMM
// 64x64 output tile per 256-thread workgroup (4 waves, one 32x32 MFMA block each), 16-deep k slabs staged through LDS in
// k-major order.  A(m,k) = A[m * a_rs + k * a_cs], B(k,n) = B[k * b_rs + n * b_cs].
struct GemmF32P {
  const float* A; long a_rs, a_cs, a_bs;
  const float* B; long b_rs, b_cs, b_bs;
  float* C; long ldc, c_bs;
  const float* bias;
  int M, N, K;
  float alpha, beta;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32P p) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* A = p.A + (long)blockIdx.z * p.a_bs;
  const float* B = p.B + (long)blockIdx.z * p.b_bs;
  float* C = p.C + (long)blockIdx.z * p.c_bs;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16 acc = f32x16{0};
  // staging maps: k fastest when k is the contiguous index of the operand, else the tile's other index fastest
  const bool a_kfast = p.a_cs == 1, b_kfast = p.b_rs == 1;
  for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int mm, kk;
      if (a_kfast) { kk = tid & 15; mm = (tid >> 4) + 16 * j; } else { mm = tid & 63; kk = (tid >> 6) + 4 * j; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < p.M && gk < p.K) ? A[(long)gm * p.a_rs + (long)gk * p.a_cs] : 0.f;
      int nn, kb;
      if (b_kfast) { kb = tid & 15; nn = (tid >> 4) + 16 * j; } else { nn = tid & 63; kb = (tid >> 6) + 4 * j; }
      const int gn = n0 + nn, gkb = k0 + kb;
      Bs[kb][nn] = (gn < p.N && gkb < p.K) ? B[(long)gkb * p.b_rs + (long)gn * p.b_cs] : 0.f;
    }
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[ks + h][wm + r], Bs[ks + h][wn + r], acc, 0, 0, 0);
    __syncthreads();
  }
  // accumulator layout (32x32): element i of lane (r, h) = C[row 8*(i/4) + 4*h + i%4][col r]
  const int r = lane & 31, h = lane >> 5;
  const int n = n0 + wn + r;
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + wm + 8 * (i >> 2) + 4 * h + (i & 3);
    if (m >= p.M) continue;
    float* c = C + (long)m * p.ldc + n;
    float v = p.alpha * acc[i] + bv;
    if (p.beta != 0.f) v += p.beta * *c;
    *c = v;
  }
}

extern "C" int wft_gemm_f32(const wft_gemm_f32_args* a, void* stream) {
  WFT_CHECK_ARG(a && a->A && a->B && a->C, "null pointer");
  WFT_CHECK_ARG(a->M >= 1 && a->N >= 1 && a->K >= 1 && a->batch >= 1 && a->batch <= 65535, "bad shape");
  WFT_CHECK_ARG(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "dims exceed int32");
  WFT_CHECK_ARG((a->M + 63) / 64 <= 65535, "M too large for one launch (tile rows > 65535)");
  GemmF32P p;
  p.A = a->A; p.a_rs = a->a_rs; p.a_cs = a->a_cs; p.a_bs = a->a_bs;
  p.B = a->B; p.b_rs = a->b_rs; p.b_cs = a->b_cs; p.b_bs = a->b_bs;
  p.C = a->C; p.ldc = a->ldc; p.c_bs = a->c_bs;
  p.bias = a->bias;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.alpha = a->alpha; p.beta = a->beta;
  dim3 grid((unsigned)((a->N + 63) / 64), (unsigned)((a->M + 63) / 64), (unsigned)a->batch);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------------------------------------ softmax
// One wave per row (fp32 online-free two-pass: rows are at most a few thousand wide and sit in L2).  Row r of the
// [nrows, cols] matrix belongs to query q = r % rows_per_mat; with `causal`, columns > q are masked (decoder self-attention:
// the reference's -inf upper-triangular mask buffer).
__global__ __launch_bounds__(256) void softmax_fwd_f32_kernel(float* s, long nrows, int cols, long ld, float scale, int causal,
                                                               int rows_per_mat) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  float* x = s + row * ld;
  const int lim = causal ? (int)(row % rows_per_mat) + 1 : cols;
  float m = -3.0e38f;
  for (int c = lane; c < lim; c += 64) m = fmaxf(m, x[c] * scale);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < lim; c += 64) sum += expf(x[c] * scale - m);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) x[c] = c < lim ? expf(x[c] * scale - m) * inv : 0.f;
}
extern "C" int wft_softmax_fwd_f32(float* s, int64_t nrows, int64_t cols, int64_t ld, float scale, int causal,
                                   int64_t rows_per_mat, void* stream) {
  WFT_CHECK_ARG(s && nrows >= 1 && cols >= 1 && ld >= cols && rows_per_mat >= 1, "bad arguments");
  hipLaunchKernelGGL(softmax_fwd_f32_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, (long)nrows,
                     (int)cols, (long)ld, scale, causal, (int)rows_per_mat);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
// ds = scale * p * (dp - sum_c p_c dp_c), written over dp
__global__ __launch_bounds__(256) void softmax_bwd_f32_kernel(const float* p, float* dp, long nrows, int cols, long ld, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  const float* pr = p + row * ld;
  float* d = dp + row * ld;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += pr[c] * d[c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) d[c] = scale * pr[c] * (d[c] - dot);
}
extern "C" int wft_softmax_bwd_f32(const float* p, float* dp, int64_t nrows, int64_t cols, int64_t ld, float scale, void* stream) {
  WFT_CHECK_ARG(p && dp && nrows >= 1 && cols >= 1 && ld >= cols, "bad arguments");
  hipLaunchKernelGGL(softmax_bwd_f32_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, (long)nrows,
                     (int)cols, (long)ld, scale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// mask = {rows_per_batch, t0, t1, c0, c1} (deep SpecAugment, model/model_utils.py:409-417) or rows_per_batch = 0
__device__ __forceinline__ bool ln_masked(long row, int c, int rpb, int t0, int t1, int c0, int c1) {
  if (rpb <= 0) return false;
  const int t = (int)(row % rpb);
  return (t >= t0 && t < t1) || (c >= c0 && c < c1);
}
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                                          float* rstd, long rows, int cols, float eps, int rpb, int t0, int t1, int c0,
                                                          int c1) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) { const float d = xr[c] - mu; v += d * d; }
  const float rs = 1.0f / sqrtf(wave_sum(v) / (float)cols + eps);
  float* yr = y + row * cols;
  for (int c = lane; c < cols; c += 64)
    yr[c] = ln_masked(row, c, rpb, t0, t1, c0, c1) ? 0.f : (xr[c] - mu) * rs * gamma[c] + beta[c];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
extern "C" int wft_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                     int64_t rows, int cols, float eps, const int32_t* mask, void* stream) {
  WFT_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows >= 1 && cols >= 1, "bad arguments");
  const int rpb = mask ? mask[0] : 0;
  hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd,
                     (long)rows, cols, eps, rpb, mask ? mask[1] : 0, mask ? mask[2] : 0, mask ? mask[3] : 0, mask ? mask[4] : 0);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
// dx of one row; the masked positions carry no gradient.  dxhat = dy * gamma.
__global__ __launch_bounds__(256) void ln_bwd_dx_f32_kernel(const float* dy, const float* x, const float* gamma, const float* mean,
                                                             const float* rstd, float* dx, long rows, int cols, int rpb, int t0, int t1,
                                                             int c0, int c1) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* dyr = dy + row * cols;
  const float* xr = x + row * cols;
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float g = ln_masked(row, c, rpb, t0, t1, c0, c1) ? 0.f : dyr[c] * gamma[c];
    const float xh = (xr[c] - mu) * rs;
    s1 += g;
    s2 += g * xh;
  }
  s1 = wave_sum(s1) / (float)cols;
  s2 = wave_sum(s2) / (float)cols;
  float* dxr = dx + row * cols;
  for (int c = lane; c < cols; c += 64) {
    const float g = ln_masked(row, c, rpb, t0, t1, c0, c1) ? 0.f : dyr[c] * gamma[c];
    const float xh = (xr[c] - mu) * rs;
    dxr[c] = rs * (g - s1 - xh * s2);
  }
}
// dgamma[c] = sum_r dy xhat, dbeta[c] = sum_r dy (masked positions excluded); one thread per column, rows in order
__global__ __launch_bounds__(64) void ln_bwd_dgb_f32_kernel(const float* dy, const float* x, const float* mean, const float* rstd,
                                                             float* dgamma, float* dbeta, long rows, int cols, int rpb, int t0, int t1,
                                                             int c0, int c1) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cols) return;
  float dg = 0.f, db = 0.f;
  for (long r = 0; r < rows; ++r) {
    if (ln_masked(r, c, rpb, t0, t1, c0, c1)) continue;
    const float g = dy[r * cols + c];
    dg += g * (x[r * cols + c] - mean[r]) * rstd[r];
    db += g;
  }
  dgamma[c] = dg;
  dbeta[c] = db;
}
extern "C" int wft_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     float* dx, float* dgamma, float* dbeta, int64_t rows, int cols, const int32_t* mask, void* stream) {
  WFT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && rows >= 1 && cols >= 1, "bad arguments");
  const int rpb = mask ? mask[0] : 0, t0 = mask ? mask[1] : 0, t1 = mask ? mask[2] : 0, c0 = mask ? mask[3] : 0, c1 = mask ? mask[4] : 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_bwd_dx_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx, (long)rows,
                     cols, rpb, t0, t1, c0, c1);
  hipLaunchKernelGGL(ln_bwd_dgb_f32_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, s, dy, x, mean, rstd, dgamma, dbeta,
                     (long)rows, cols, rpb, t0, t1, c0, c1);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void gelu_fwd_f32_kernel(const float* x, float* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // exact-erf GELU (torch.nn.GELU default)
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_f32_kernel(const float* dy, const float* x, float* dx, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    dx[i] = dy[i] * (cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v));
  }
}
__global__ __launch_bounds__(256) void axpby_f32_kernel(float a, const float* x, float b, const float* y, float* out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}
static unsigned ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g > 65535 ? 65535 : g);
}
extern "C" int wft_gelu_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
  WFT_CHECK_ARG(x && y && n >= 1, "bad arguments");
  hipLaunchKernelGGL(gelu_fwd_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
extern "C" int wft_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  WFT_CHECK_ARG(dy && x && dx && n >= 1, "bad arguments");
  hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
extern "C" int wft_axpby_f32(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream) {
  WFT_CHECK_ARG(x && out && n >= 1, "bad arguments");
  hipLaunchKernelGGL(axpby_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, x, b, y, out, (long)n);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
// out[c] = sum_r x[r, c] (bias gradients), rows added in order: one thread per column
__global__ __launch_bounds__(64) void colsum_f32_kernel(const float* x, long rows, int cols, long ld, float* out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (long r = 0; r < rows; ++r) s += x[r * ld + c];
  out[c] = s;
}
extern "C" int wft_colsum_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out, void* stream) {
  WFT_CHECK_ARG(x && out && rows >= 1 && cols >= 1 && ld >= cols, "bad arguments");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, (long)rows, (int)cols,
                     (long)ld, out);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embed_fwd_f32_kernel(const long* tokens, const float* emb, const float* pos, float* out, long S,
                                                             int d) {
  const long row = blockIdx.x;  // b * S + s
  const long tok = tokens[row];
  const float* e = emb + tok * d;
  const float* pp = pos + (row % S) * d;
  for (int c = threadIdx.x; c < d; c += 256) out[row * d + c] = e[c] + pp[c];
}
extern "C" int wft_embed_fwd_f32(const int64_t* tokens, const float* emb, const float* pos, float* out, int64_t B, int64_t S, int d,
                                 void* stream) {
  WFT_CHECK_ARG(tokens && emb && pos && out && B >= 1 && S >= 1 && d >= 1, "bad arguments");
  hipLaunchKernelGGL(embed_fwd_f32_kernel, dim3((unsigned)(B * S)), dim3(256), 0, (hipStream_t)stream, (const long*)tokens, emb, pos, out,
                     (long)S, d);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
// demb: one workgroup per vocabulary row THAT OCCURS is found by scanning the (short) token list in order: deterministic,
// no atomics.  grid = distinct handling by token position: block i handles token i only if no earlier position holds the
// same id, and then adds every position with that id in order.  dpos[s] = sum_b dout[b, s].
__global__ __launch_bounds__(256) void embed_bwd_f32_kernel(const long* tokens, const float* dout, float* demb, float* dpos, long B, long S,
                                                             int d) {
  const long n = B * S;
  const long i = blockIdx.x;
  if (i < n) {
    const long tok = tokens[i];
    for (long j = 0; j < i; ++j)
      if (tokens[j] == tok) return;  // an earlier block owns this id
    for (int c = threadIdx.x; c < d; c += 256) {
      float s = 0.f;
      for (long j = i; j < n; ++j)
        if (tokens[j] == tok) s += dout[j * d + c];
      demb[tok * d + c] = s;  // demb was zero-filled by the caller
    }
  } else {
    const long spos = i - n;
    for (int c = threadIdx.x; c < d; c += 256) {
      float s = 0.f;
      for (long b = 0; b < B; ++b) s += dout[(b * S + spos) * d + c];
      dpos[spos * d + c] = s;
    }
  }
}
extern "C" int wft_embed_bwd_f32(const int64_t* tokens, const float* dout, float* demb, float* dpos, int64_t B, int64_t S, int d,
                                 void* stream) {
  WFT_CHECK_ARG(tokens && dout && demb && dpos && B >= 1 && S >= 1 && d >= 1, "bad arguments");
  hipLaunchKernelGGL(embed_bwd_f32_kernel, dim3((unsigned)(B * S + S)), dim3(256), 0, (hipStream_t)stream, (const long*)tokens, dout, demb,
                     dpos, (long)B, (long)S, d);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------------------------------------ cross entropy
// F.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing) on fp32 logits [rows, V] (model/model_utils.py:66):
// row_stats[row] = {loss_row (0 if ignored), lse}; stats = {sum of row losses, number of non-ignored rows}.
__global__ __launch_bounds__(256) void ce_fwd_f32_kernel(const float* logits, long ld, const long* targets, long V, float eps,
                                                          float* row_loss, float* row_lse) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* x = logits + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float m = -3.0e38f;
  for (long c = tid; c < V; c += 256) m = fmaxf(m, x[c]);
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f, sx = 0.f;
  for (long c = tid; c < V; c += 256) { s += expf(x[c] - m); sx += x[c]; }
  s = wave_sum(s);
  sx = wave_sum(sx);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  const float S = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  if (lane == 0) red[wv] = sx;
  __syncthreads();
  if (tid == 0) {
    const float X = (red[0] + red[1]) + (red[2] + red[3]);
    const float lse = m + logf(S);
    const long t = targets[row];
    row_loss[row] = (t >= 0 && t < V) ? (1.f - eps) * (lse - x[t]) + eps * (lse - X / (float)V) : 0.f;
    row_lse[row] = lse;
  }
}
__global__ __launch_bounds__(256) void ce_reduce_f32_kernel(const float* row_loss, const long* targets, long rows, long V, float* stats) {
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
extern "C" int wft_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V, float label_smoothing,
                              float* row_loss, float* row_lse, float* stats, void* stream) {
  WFT_CHECK_ARG(logits && targets && row_loss && row_lse && stats && rows >= 1 && V >= 1 && ld >= V, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_f32_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, (long)ld, (const long*)targets, (long)V,
                     label_smoothing, row_loss, row_lse);
  hipLaunchKernelGGL(ce_reduce_f32_kernel, dim3(1), dim3(256), 0, s, row_loss, (const long*)targets, (long)rows, (long)V, stats);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
// dlogits = gscale / n_valid * (softmax - (1 - eps) onehot - eps / V) on valid rows, 0 elsewhere (written over the logits)
__global__ __launch_bounds__(256) void ce_bwd_f32_kernel(float* logits, long ld, const long* targets, long V, float eps,
                                                          const float* row_lse, const float* stats, const float* gscale) {
  const long row = blockIdx.x;
  float* x = logits + row * ld;
  const long t = targets[row];
  const bool valid = t >= 0 && t < V;
  const float coef = valid ? gscale[0] / fmaxf(stats[1], 1.f) : 0.f;
  const float lse = row_lse[row], sm = eps / (float)V;
  for (long c = threadIdx.x; c < V; c += 256) {
    float g = expf(x[c] - lse) - sm;
    if (c == t) g -= (1.f - eps);
    x[c] = g * coef;
  }
}
extern "C" int wft_ce_bwd_f32(float* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V, float label_smoothing,
                              const float* row_lse, const float* stats, const float* gscale, void* stream) {
  WFT_CHECK_ARG(logits && targets && row_lse && stats && gscale && rows >= 1 && V >= 1 && ld >= V, "bad arguments");
  hipLaunchKernelGGL(ce_bwd_f32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, (const long*)targets,
                     (long)V, label_smoothing, row_lse, stats, gscale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
