// optim.hip — optimizer step kernels: multi-tensor AdamW with the clip_grad_norm_
// coefficient folded in, and the element-wise / layout parts of Muon (the Newton-Schulz products themselves
// are wft_gemm_nt_bf16 launches, batched over all matrices of one shape).
//
// All kernels are HBM-bound streaming passes: AdamW moves 28 B per parameter (p, g, m, v read; p, m, v
// written) in ONE launch for the whole model instead of one launch per tensor, and the gradient-norm pass
// reads 4 B per parameter with a fixed-order two-stage reduction (bitwise reproducible).
//
// "Multi-tensor" = a pointer table in DEVICE memory (caller-owned, rebuilt by the host whenever a pointer
// changes): tab[row * n + t] is the address of tensor t's row-th array; numel[t] its element count;
// chunk_start[t] the index of its first WFT_MT_CHUNK-element chunk (chunk_start[n] = total chunks).
// One workgroup processes one chunk and finds its tensor by binary search over chunk_start.
#include "common.h"

#define MT_CHUNK WFT_MT_CHUNK

__device__ __forceinline__ int mt_find(const int* chunk_start, int n, int chunk) {
  int lo = 0, hi = n;  // invariant: chunk_start[lo] <= chunk < chunk_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float block_sum_256(float s, float* red) {
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ gradient norm (clip_grad_norm_, model_utils.py:107)
__global__ __launch_bounds__(256) void mt_sumsq_kernel(const long* tab, const long* numel, const int* chunk_start, int n,
                                                       float* partial) {
  __shared__ float red[4];
  const int t = mt_find(chunk_start, n, blockIdx.x);
  if (tab[t] == 0) {  // no gradient this step (workgroup-uniform): contributes nothing
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.f;
    return;
  }
  const long off = (long)(blockIdx.x - chunk_start[t]) * MT_CHUNK;
  const long cnt = numel[t] - off < MT_CHUNK ? numel[t] - off : MT_CHUNK;
  const float* g = (const float*)tab[t] + off;
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    const long nv = cnt >> 2;
    for (long i = threadIdx.x; i < nv; i += 256) {
      const f32x4 x = *(const f32x4*)(g + i * 4);
      s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    for (long i = (nv << 2) + threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
  } else {
    for (long i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// fixed-order sum of the per-chunk partials (one workgroup)
__global__ __launch_bounds__(256) void mt_sumsq_final_kernel(const float* partial, int total, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < total; i += 256) s += partial[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

extern "C" int wft_mt_sumsq_f32(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n, int total_chunks,
                                float* partial, float* out, void* stream) {
  WFT_CHECK_ARG(tab && numel && chunk_start && partial && out && n >= 1 && total_chunks >= 1, "bad args");
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)tab,
                     (const long*)numel, (const int*)chunk_start, n, partial);
  hipLaunchKernelGGL(mt_sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, total_chunks, out);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW semantics)
// tab rows: 0 p, 1 g, 2 exp_avg, 3 exp_avg_sq.  If sumsq != NULL the gradient is first scaled by
// min(1, max_norm / (sqrt(*sumsq) + 1e-6)) — exactly torch.nn.utils.clip_grad_norm_'s coefficient.
__global__ __launch_bounds__(256) void mt_adamw_kernel(const long* tab, const long* numel, const int* chunk_start, int n,
                                                       float lr, float b1, float b2, float eps, float wd, float bc1,
                                                       float bc2, const float* sumsq, float max_norm) {
  const int t = mt_find(chunk_start, n, blockIdx.x);
  const long off = (long)(blockIdx.x - chunk_start[t]) * MT_CHUNK;
  const long cnt = numel[t] - off < MT_CHUNK ? numel[t] - off : MT_CHUNK;
  float* p = (float*)tab[t] + off;
  const float* g = (const float*)tab[n + t] + off;
  float* m = (float*)tab[2 * n + t] + off;
  float* v = (float*)tab[3 * n + t] + off;
  float gs = 1.f;
  if (sumsq) {
    const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    gs = coef < 1.f ? coef : 1.f;
  }
  const float step = lr / bc1, rbc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= gs;
    pp *= decay;
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    pp -= step * mm / (sqrtf(vv) * rbc2 + eps);
  };
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const long nv = al ? cnt >> 2 : 0;
  for (long i = threadIdx.x; i < nv; i += 256) {
    f32x4 pp = *(f32x4*)(p + i * 4), mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
    const f32x4 gg = *(const f32x4*)(g + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      upd(pe, gg[e], me, ve);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    *(f32x4*)(p + i * 4) = pp;
    *(f32x4*)(m + i * 4) = mm;
    *(f32x4*)(v + i * 4) = vv;
  }
  for (long i = (nv << 2) + threadIdx.x; i < cnt; i += 256) upd(p[i], g[i], m[i], v[i]);
}

extern "C" int wft_mt_adamw(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n, int total_chunks,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                            float bias_corr2, const float* sumsq, float max_norm, void* stream) {
  WFT_CHECK_ARG(tab && numel && chunk_start && n >= 1 && total_chunks >= 1, "bad args");
  WFT_CHECK_ARG(!sumsq || max_norm > 0.f, "max_norm must be > 0 when a gradient norm is given");
  hipLaunchKernelGGL(mt_adamw_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)tab,
                     (const long*)numel, (const int*)chunk_start, n, lr, beta1, beta2, eps, weight_decay, bias_corr1,
                     bias_corr2, sumsq, max_norm);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------ 8-bit AdamW (bnb.optim.AdamW8bit; reference model/optimizer.py:241-256)
// Block-wise dynamic-quantised moments: one byte per element, blocks of 2 048 elements with an f32 absmax each (include/wft.h;
// restated in oracle/adam8bit_oracle.py — the package is not in the image: parity unpinned).  HBM-bound: 16 B per parameter
// (p read + written, g read, two code bytes read + written).  One workgroup walks the up to 32 blocks of a chunk; a thread owns
// 8 consecutive elements of a block (two 16-byte loads of p and g, one 8-byte load of each code array).  Both maps live in LDS:
// decoding is one LDS read per code; encoding is an 8-step bisection over the 255 midpoints between neighbouring map entries
// (code = number of midpoints below the normalised value: the nearest entry, ties to the lower code).
__device__ __forceinline__ unsigned q8_encode(const float* mid, float x) {
  // mid[0..254] ascending; invariant: mid[lo - 1] < x (or lo == 0), and x <= mid[hi] (or hi == 255)
  unsigned lo = 0;
#pragma unroll
  for (unsigned step = 128; step >= 1; step >>= 1) {
    const unsigned probe = lo + step;  // candidate code: valid if mid[probe - 1] < x
    if (probe <= 255 && mid[probe - 1] < x) lo = probe;
  }
  return lo;
}

__global__ __launch_bounds__(256) void mt_adamw8_kernel(const long* tab, const long* numel, const int* chunk_start, int n,
                                                        const float* qmap1, const float* qmap2, float lr, float b1, float b2,
                                                        float eps, float wd, float bc1, float bc2, const float* sumsq,
                                                        float max_norm) {
  __shared__ float q1[256], q2[256], mid1[256], mid2[256];
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  q1[tid] = qmap1[tid];
  q2[tid] = qmap2[tid];
  if (tid < 255) {
    mid1[tid] = (qmap1[tid] + qmap1[tid + 1]) * 0.5f;
    mid2[tid] = (qmap2[tid] + qmap2[tid + 1]) * 0.5f;
  }
  const int t = mt_find(chunk_start, n, blockIdx.x);
  const long off = (long)(blockIdx.x - chunk_start[t]) * MT_CHUNK;
  const long cnt = numel[t] - off < MT_CHUNK ? numel[t] - off : MT_CHUNK;
  float* p = (float*)tab[t] + off;
  const float* g = (const float*)tab[n + t] + off;
  unsigned char* s1 = (unsigned char*)tab[2 * n + t] + off;
  unsigned char* s2 = (unsigned char*)tab[3 * n + t] + off;
  float* am1 = (float*)tab[4 * n + t] + (off / WFT_Q8_BLOCK);
  float* am2 = (float*)tab[5 * n + t] + (off / WFT_Q8_BLOCK);
  float gs = 1.f;
  if (sumsq) {
    const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    gs = coef < 1.f ? coef : 1.f;
  }
  const float c2 = sqrtf(bc2);
  const float step_size = -lr * c2 / bc1, eps2 = c2 * eps, decay = 1.f - lr * wd;
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g)) & 15) == 0 && ((((uintptr_t)s1) | ((uintptr_t)s2)) & 7) == 0;
  __syncthreads();
  const int nblk = (int)((cnt + WFT_Q8_BLOCK - 1) / WFT_Q8_BLOCK);
  for (int b = 0; b < nblk; ++b) {
    const long e0 = (long)b * WFT_Q8_BLOCK + tid * 8;
    const int nvalid = (int)(cnt - e0 < 8 ? (cnt - e0 < 0 ? 0 : cnt - e0) : 8);
    float pv[8], gv[8], m[8], v[8];
    unsigned char c1[8], c2b[8];
    if (al && nvalid == 8) {
      *(f32x4*)pv = *(const f32x4*)(p + e0);
      *(f32x4*)(pv + 4) = *(const f32x4*)(p + e0 + 4);
      *(f32x4*)gv = *(const f32x4*)(g + e0);
      *(f32x4*)(gv + 4) = *(const f32x4*)(g + e0 + 4);
      *(u32x2*)c1 = *(const u32x2*)(s1 + e0);
      *(u32x2*)c2b = *(const u32x2*)(s2 + e0);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = e < nvalid;
        pv[e] = ok ? p[e0 + e] : 0.f;
        gv[e] = ok ? g[e0 + e] : 0.f;
        c1[e] = ok ? s1[e0 + e] : 0;
        c2b[e] = ok ? s2[e0 + e] : 0;
      }
    }
    const float a1 = am1[b], a2 = am2[b];
    float mx1 = 0.f, mx2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gg = gv[e] * gs;
      m[e] = q1[c1[e]] * a1 * b1 + (1.f - b1) * gg;
      v[e] = q2[c2b[e]] * a2 * b2 + (1.f - b2) * gg * gg;
      if (e >= nvalid) { m[e] = 0.f; v[e] = 0.f; }
      mx1 = fmaxf(mx1, fabsf(m[e]));
      mx2 = fmaxf(mx2, v[e]);
    }
    mx1 = wave_max(mx1);
    mx2 = wave_max(mx2);
    __syncthreads();  // (the previous block's readers of `red` are done)
    if ((tid & 63) == 0) { red[0][tid >> 6] = mx1; red[1][tid >> 6] = mx2; }
    __syncthreads();
    const float n1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float n2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const float i1 = n1 > 0.f ? 1.f / n1 : 0.f, i2 = n2 > 0.f ? 1.f / n2 : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pv[e] = pv[e] + step_size * (m[e] / (sqrtf(v[e]) + eps2));
      if (wd > 0.f) pv[e] *= decay;
      c1[e] = (unsigned char)q8_encode(mid1, m[e] * i1);
      c2b[e] = (unsigned char)q8_encode(mid2, v[e] * i2);
    }
    if (al && nvalid == 8) {
      *(f32x4*)(p + e0) = *(const f32x4*)pv;
      *(f32x4*)(p + e0 + 4) = *(const f32x4*)(pv + 4);
      *(u32x2*)(s1 + e0) = *(const u32x2*)c1;
      *(u32x2*)(s2 + e0) = *(const u32x2*)c2b;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nvalid) { p[e0 + e] = pv[e]; s1[e0 + e] = c1[e]; s2[e0 + e] = c2b[e]; }
    }
    if (tid == 0) { am1[b] = n1; am2[b] = n2; }
  }
}

extern "C" int wft_mt_adamw8(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n, int total_chunks,
                             const float* qmap1, const float* qmap2, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bias_corr1, float bias_corr2, const float* sumsq, float max_norm, void* stream) {
  WFT_CHECK_ARG(tab && numel && chunk_start && qmap1 && qmap2 && n >= 1 && total_chunks >= 1, "bad args");
  WFT_CHECK_ARG(!sumsq || max_norm > 0.f, "max_norm must be > 0 when a gradient norm is given");
  WFT_CHECK_ARG(bias_corr1 > 0.f && bias_corr2 > 0.f, "bias corrections must be > 0");
  hipLaunchKernelGGL(mt_adamw8_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)tab, (const long*)numel,
                     (const int*)chunk_start, n, qmap1, qmap2, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, sumsq,
                     max_norm);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// ------------------------------------------------------------------ Muon (muon.py: muon_update, zeropower_via_newtonschulz5)
// Step 1, per matrix t of a same-shape group (tab rows: 0 p (unused here), 1 g, 2 momentum buffer):
//   buf = lerp(buf, g, 1 - beta);  u = nesterov ? lerp(g, buf, beta) : buf      (g is overwritten with u, as grad.lerp_ does)
//   U[t] = bf16(u) in the parameter's own [rows, cols] orientation; partial[t][chunk] = sum bf16(u)^2 (fp32)
// grid = (chunks_per_mat, n_mats); every matrix has `numel` elements.
__global__ __launch_bounds__(256) void muon_momentum_kernel(const long* tab, int n, long numel, float beta, int nesterov,
                                                            unsigned short* U, float* partial, const float* sumsq,
                                                            float max_norm) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  const long off = (long)blockIdx.x * MT_CHUNK;
  const long cnt = numel - off < MT_CHUNK ? numel - off : MT_CHUNK;
  // a NULL gradient row is a parameter that took no gradient this step (a layer stochastic depth skipped): the package gives it
  // a zero gradient (muon.py "force synchronization"), i.e. the momentum decays and the update comes from the momentum alone —
  // computed here without materialising the zeros
  float* g = tab[n + t] ? (float*)tab[n + t] + off : nullptr;
  float* buf = (float*)tab[2 * n + t] + off;
  unsigned short* u = U + (long)t * numel + off;
  float gs = 1.f;
  if (sumsq) {
    const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    gs = coef < 1.f ? coef : 1.f;
  }
  float s = 0.f;
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const float gg = g ? g[i] * gs : 0.f;
    const float bb = buf[i] + (1.f - beta) * (gg - buf[i]);  // torch lerp: start + weight * (end - start)
    const float uu = nesterov ? gg + beta * (bb - gg) : bb;
    buf[i] = bb;
    if (g) g[i] = uu;
    const unsigned short ub = f2bf(uu);
    u[i] = ub;
    const float ur = bf2f(ub);
    s += ur * ur;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[(long)t * gridDim.x + blockIdx.x] = s;
}

extern "C" int wft_muon_momentum_mt(const void* tab, int n_mats, int64_t numel, float beta, int nesterov, wft_bf16* U,
                                    float* partial, const float* sumsq, float max_norm, void* stream) {
  WFT_CHECK_ARG(tab && U && partial && n_mats >= 1 && numel >= 1, "bad args");
  const int chunks = (int)cdiv64(numel, MT_CHUNK);
  WFT_CHECK_ARG(n_mats <= 65535, "at most 65535 matrices per group");
  hipLaunchKernelGGL(muon_momentum_kernel, dim3(chunks, n_mats), dim3(256), 0, (hipStream_t)stream, (const long*)tab, n_mats,
                     (long)numel, beta, nesterov, U, partial, sumsq, max_norm);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// Step 2: X = U / (||U||_F + 1e-7) in bf16 arithmetic (norm rounded to bf16 first, as X.norm() of a bf16 tensor is),
// written in BOTH orientations with zero padding: X [n][Rp][Cp] has rows <= cols (it is U^T when the parameter is
// tall), Xt [n][Cp][Rp] is its transpose.  64x64 tiles through LDS; grid = (tiles_c, tiles_r, n_mats) over the
// PADDED [Rp, Cp] frame so that the pad is written (zeros) too.
__global__ __launch_bounds__(256) void muon_prepare_kernel(const unsigned short* U, int rows, int cols, int transpose,
                                                           const float* partial, int chunks, unsigned short* X,
                                                           unsigned short* Xt, int Rp, int Cp) {
  __shared__ unsigned short tile[64][66];
  __shared__ float red[4];
  const int t = blockIdx.z;
  float s = 0.f;
  for (int i = threadIdx.x; i < chunks; i += 256) s += partial[(long)t * chunks + i];
  s = block_sum_256(s, red);
  const float nrm = bf2f(f2bf(sqrtf(s))) + 1e-7f;
  const int R = transpose ? cols : rows, C = transpose ? rows : cols;  // logical X is [R, C]
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const unsigned short* u = U + (long)t * rows * cols;
  // load X[r0.., c0..] into tile[r][c]
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int a, b;  // (a, b): coalesced along U's contiguous dimension
    if (!transpose) { a = i >> 6; b = i & 63; } else { b = i >> 6; a = i & 63; }
    const int r = r0 + a, c = c0 + b;
    unsigned short val = 0;
    if (r < R && c < C) {
      const float x = bf2f(transpose ? u[(long)c * cols + r] : u[(long)r * cols + c]);
      val = f2bf(x / nrm);
    }
    tile[a][b] = val;
  }
  __syncthreads();
  unsigned short* x = X + (long)t * Rp * Cp;
  unsigned short* xt = Xt + (long)t * Cp * Rp;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int a = i >> 6, b = i & 63;
    if (r0 + a < Rp && c0 + b < Cp) x[(long)(r0 + a) * Cp + c0 + b] = tile[a][b];
    if (c0 + a < Cp && r0 + b < Rp) xt[(long)(c0 + a) * Rp + r0 + b] = tile[b][a];
  }
}

extern "C" int wft_muon_prepare(const wft_bf16* U, int rows, int cols, const float* partial, int chunks, wft_bf16* X,
                                wft_bf16* Xt, int rows_pad, int cols_pad, int n_mats, void* stream) {
  WFT_CHECK_ARG(U && partial && X && Xt && rows >= 1 && cols >= 1 && n_mats >= 1 && chunks >= 1, "bad args");
  const int transpose = rows > cols;
  const int R = transpose ? cols : rows, C = transpose ? rows : cols;
  WFT_CHECK_ARG(rows_pad >= R && cols_pad >= C, "pads are for the rows<=cols orientation: rows_pad >= min(rows, cols), cols_pad >= max");
  hipLaunchKernelGGL(muon_prepare_kernel, dim3((cols_pad + 63) / 64, (rows_pad + 63) / 64, n_mats), dim3(256), 0,
                     (hipStream_t)stream, U, rows, cols, transpose, partial, chunks, X, Xt, rows_pad, cols_pad);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// batched bf16 transpose: dst[b][c][r] = src[b][r][c]
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* src, int rows, int cols, unsigned short* dst) {
  __shared__ unsigned short tile[64][66];
  const long boff = (long)blockIdx.z * rows * cols;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int a = i >> 6, b = i & 63;
    tile[a][b] = (r0 + a < rows && c0 + b < cols) ? src[boff + (long)(r0 + a) * cols + c0 + b] : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int a = i >> 6, b = i & 63;
    if (c0 + a < cols && r0 + b < rows) dst[boff + (long)(c0 + a) * rows + r0 + b] = tile[b][a];
  }
}

extern "C" int wft_transpose_bf16(const wft_bf16* src, int rows, int cols, wft_bf16* dst, int batch, void* stream) {
  WFT_CHECK_ARG(src && dst && rows >= 1 && cols >= 1 && batch >= 1 && batch <= 65535, "bad args");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                     src, rows, cols, dst);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// Step 4: p = p * (1 - lr*wd) - lr * scale * O[t][r][c]   (O: bf16 Newton-Schulz result in p's orientation, row stride ldo,
// matrix stride so; scale = sqrt(max(1, rows/cols)), muon.py muon_update).  grid = (chunks, n_mats).
__global__ __launch_bounds__(256) void muon_apply_kernel(const long* tab, int n, int rows, int cols, const unsigned short* O,
                                                         long ldo, long so, float lr, float wd, float scale) {
  const int t = blockIdx.y;
  const long numel = (long)rows * cols;
  const long off = (long)blockIdx.x * MT_CHUNK;
  const long cnt = numel - off < MT_CHUNK ? numel - off : MT_CHUNK;
  float* p = (float*)tab[t];
  const unsigned short* o = O + (long)t * so;
  const float decay = 1.f - lr * wd, a = lr * scale;
  for (long i = off + threadIdx.x; i < off + cnt; i += 256) {
    const long r = i / cols, c = i - r * cols;
    p[i] = p[i] * decay - a * bf2f(o[r * ldo + c]);
  }
}

extern "C" int wft_muon_apply_mt(const void* tab, int n_mats, int rows, int cols, const wft_bf16* O, int64_t ldo,
                                 int64_t stride_o, float lr, float weight_decay, float scale, void* stream) {
  WFT_CHECK_ARG(tab && O && n_mats >= 1 && n_mats <= 65535 && rows >= 1 && cols >= 1 && ldo >= cols, "bad args");
  const int chunks = (int)cdiv64((int64_t)rows * cols, MT_CHUNK);
  hipLaunchKernelGGL(muon_apply_kernel, dim3(chunks, n_mats), dim3(256), 0, (hipStream_t)stream, (const long*)tab, n_mats, rows,
                     cols, O, (long)ldo, (long)stride_o, lr, weight_decay, scale);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
