// audio.hip — log-mel front end and SpecAugment on the GPU.
//
// logmel: one 256-thread workgroup per (clip, 16 frames): the 2800-sample reflect-padded
// audio span is staged once in LDS, Hann-windowed frames are written to LDS and folded on the DFT's
// symmetry (x[n] +- x[400 - n]: half the multiply-adds), the 400-point DFT is evaluated directly in fp32
// from a 400-entry twiddle table in LDS (exact periodic index k*n mod 400, no trig in the loop), then the [16 x 201] power tile is multiplied by
// the mel filterbank and log10'd.  Algorithmic HBM bytes per clip: 1.92 MB read +
// n_mels*3000*4 B written twice (the clip-max floor needs a second pass).
// specaug: one pass, bilinear time-warp gather + time/frequency/extremes masks.
#include "common.h"

#define NFFT 400
#define HOP 160
#define NBIN 201
#define FPB 16  // frames per block

__global__ __launch_bounds__(256) void logmel_kernel(const float* audio, const float* filters, float* out,
                                                      unsigned int* clipmax, int n_samples, int n_mels, int n_frames) {
  __shared__ float span[(FPB - 1) * HOP + NFFT];  // 2800
  __shared__ __attribute__((aligned(16))) float frames[FPB][NFFT];  // windowed
  __shared__ __attribute__((aligned(8))) float tw[NFFT][2];         // (cos, sin)(2 pi i / 400)
  __shared__ float power[FPB][NBIN + 3];
  __shared__ float wmax[4];
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * FPB;
  const int b = blockIdx.y;
  const float* a = audio + (long)b * n_samples;
  for (int i = tid; i < (FPB - 1) * HOP + NFFT; i += 256) {
    int idx = f0 * HOP + i - NFFT / 2;  // position in the un-padded clip
    if (idx < 0) idx = -idx;
    if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
    idx = idx < 0 ? 0 : (idx >= n_samples ? n_samples - 1 : idx);
    span[i] = a[idx];
  }
  for (int i = tid; i < NFFT; i += 256) {
    // sincospi keeps the table accurate to 1 ulp: angle = 2*pi*i/400
    float sn, cs;
    sincospif((float)i / 200.0f, &sn, &cs);
    tw[i][0] = cs;
    tw[i][1] = sn;
  }
  __syncthreads();
  for (int i = tid; i < FPB * NFFT; i += 256) {
    const int f = i / NFFT, n = i - f * NFFT;
    const float w = 0.5f - 0.5f * tw[n][0];  // periodic Hann
    frames[f][n] = w * span[f * HOP + n];
  }
  __syncthreads();
  // Fold every windowed frame on the DFT's symmetry (round 5: half the multiply-adds).  cos(2 pi k (400 - n) / 400) =
  // cos(2 pi k n / 400) and sin(...) = -sin(...), so with s[n] = x[n] + x[400 - n], d[n] = x[n] - x[400 - n] (n = 1..199):
  //   Re X[k] = x[0] + (-1)^k x[200] + sum_n s[n] cos(2 pi k n / 400),   Im X[k] = -sum_n d[n] sin(2 pi k n / 400)
  // In place: [f][0] = x[0], [f][1..199] = s, [f][200] = x[200], [f][201..399] = d[1..199] (so that both 4-sample reads of the loop
  // below are 16-byte aligned); two phases around a barrier because d[n] lands in another pair's input.
  {
    constexpr int PAIRS = FPB * 199, PER = (PAIRS + 255) / 256;
    float pa[PER], pb[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * 256;
      if (i < PAIRS) {
        const int f = i / 199, n = 1 + (i - f * 199);
        pa[j] = frames[f][n];
        pb[j] = frames[f][NFFT - n];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * 256;
      if (i < PAIRS) {
        const int f = i / 199, n = 1 + (i - f * 199);
        frames[f][n] = pa[j] + pb[j];
        frames[f][200 + n] = pa[j] - pb[j];
      }
    }
    __syncthreads();
  }
  // DFT, register-blocked: thread = one bin k for all FPB frames.  Per 4 samples: 4 twiddle pairs (ds_read_b64) + 2 FPB
  // broadcast float4 frame reads feed 8*FPB FMAs: VALU-bound.  (The sample at offset 200 of the first d read is x[200], multiplied
  // by sin(0) = 0: an exact zero.)
  if (tid < NBIN) {
    const int k = tid;
    float re[FPB], im[FPB];
#pragma unroll
    for (int f = 0; f < FPB; ++f) { re[f] = 0.f; im[f] = 0.f; }
    int idx = 0;
    for (int n = 0; n < 200; n += 4) {
      f32x2 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = *(const f32x2*)tw[idx];
        idx += k;
        idx = idx >= NFFT ? idx - NFFT : idx;
      }
#pragma unroll
      for (int f = 0; f < FPB; ++f) {
        const f32x4 xs = *(const f32x4*)&frames[f][n];
        const f32x4 xd = *(const f32x4*)&frames[f][200 + n];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          re[f] = fmaf(xs[j], t[j][0], re[f]);
          im[f] = fmaf(xd[j], t[j][1], im[f]);
        }
      }
    }
    const float sgn = (k & 1) ? -1.f : 1.f;  // cos(pi k)
#pragma unroll
    for (int f = 0; f < FPB; ++f) {
      re[f] = fmaf(frames[f][200], sgn, re[f]);
      power[f][k] = re[f] * re[f] + im[f] * im[f];
    }
  }
  // The mel filters are triangles: 2-30 non-zero bins out of 201 per filter.  Each filter's non-zero range is found once per
  // workgroup (one thread per filter; skipping exact zeros leaves every sum bit-identical to the dense loop).
  __shared__ short mlo[256], mhi[256];
  if (tid < n_mels && tid < 256) {
    const float* fl = filters + (long)tid * NBIN;
    int lo = NBIN, hi = 0;
    for (int k = 0; k < NBIN; ++k)
      if (fl[k] != 0.f) { lo = lo < k ? lo : k; hi = k + 1; }
    mlo[tid] = (short)(lo < hi ? lo : 0);
    mhi[tid] = (short)hi;
  }
  __syncthreads();
  float lmax = -1.0e30f;
  for (int o = tid; o < n_mels * FPB; o += 256) {
    const int mI = o / FPB, f = o - mI * FPB;
    const float* fl = filters + (long)mI * NBIN;
    float acc = 0.f;
    const int k0 = mI < 256 ? mlo[mI] : 0, k1 = mI < 256 ? mhi[mI] : NBIN;
    for (int k = k0; k < k1; ++k) acc = fmaf(fl[k], power[f][k], acc);
    const float lg = log10f(fmaxf(acc, 1.0e-10f));
    const int fr = f0 + f;
    if (fr < n_frames) {
      out[((long)b * n_mels + mI) * n_frames + fr] = lg;
      lmax = fmaxf(lmax, lg);
    }
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) wmax[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // log10 values are >= -10, so (mx + 20) is a positive float: its bit pattern orders like an int
    atomicMax(clipmax + b, __builtin_bit_cast(unsigned int, mx + 20.0f));
  }
}

__global__ __launch_bounds__(256) void logmel_norm_kernel(float* out, const unsigned int* clipmax, long per_clip, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / per_clip);
    const float mx = __builtin_bit_cast(float, clipmax[b]) - 20.0f;
    const float v = fmaxf(out[i], mx - 8.0f);
    out[i] = (v + 4.0f) / 4.0f;
  }
}

extern "C" int wft_logmel(const float* audio, const float* filters, float* out, float* clipmax, int B, int n_samples,
                          int n_mels, int n_frames, void* stream) {
  WFT_CHECK_ARG(audio && filters && out && clipmax, "null pointer");
  WFT_CHECK_ARG(B >= 1 && n_mels >= 1 && n_frames >= 1 && n_samples == n_frames * HOP, "n_samples must be 160*n_frames");
  WFT_CHECK_ARG(n_samples > NFFT, "clip too short");
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(clipmax, 0, B * sizeof(float), s);
  dim3 grid((n_frames + FPB - 1) / FPB, B);
  hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, s, audio, filters, out, (unsigned int*)clipmax, n_samples, n_mels,
                     n_frames);
  const long per_clip = (long)n_mels * n_frames, total = per_clip * B;
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(logmel_norm_kernel, dim3((unsigned)g), dim3(256), 0, s, out, (const unsigned int*)clipmax, per_clip,
                     total);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// --------------------------------------------------------------------------------- SpecAugment
// params per clip: {apply_warp, warp_p, warp_d, t0, t1, f0, f1, unused}; extremes: {low_len, high_len}
__global__ __launch_bounds__(256) void specaug_kernel(const float* in, float* out, const int* params, const int* extremes,
                                                       int n_mels, int T) {
  const int b = blockIdx.z, mI = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int* pr = params + b * 8;
  const int apply_warp = pr[0], wp = pr[1], wd = pr[2], t0 = pr[3], t1 = pr[4], f0 = pr[5], f1 = pr[6];
  const int lo = extremes ? extremes[b * 2] : 0, hi = extremes ? extremes[b * 2 + 1] : 0;
  const float* row = in + ((long)b * n_mels + mI) * T;
  float v;
  if (apply_warp) {
    // cubic Hermite spline through (0,-1), (wp, y1), (T-1, 1)  (data/utils.py:66-85,110-136)
    const float L1 = (float)(T - 1);
    const float y1 = (float)(wp - wd) * 2.0f / L1 - 1.0f;
    const float sa = (y1 + 1.0f) / (float)wp;           // secant of segment 0
    const float sb = (1.0f - y1) / (float)(T - 1 - wp);  // secant of segment 1
    const float mmid = (sa + sb) * 0.5f;
    const float xs = (float)t;
    float xl, dx, yl, yr, ml, mr;
    if (t <= wp) { xl = 0.f; dx = (float)wp; yl = -1.0f; yr = y1; ml = sa; mr = mmid; }
    else { xl = (float)wp; dx = (float)(T - 1 - wp); yl = y1; yr = 1.0f; ml = mmid; mr = sb; }
    const float u = (xs - xl) / dx;
    const float u2 = u * u, u3 = u2 * u;
    const float h0 = 1.0f - 3.0f * u2 + 2.0f * u3;
    const float h1 = u - 2.0f * u2 + u3;
    const float h2 = 3.0f * u2 - 2.0f * u3;
    const float h3 = -u2 + u3;
    const float ys = h0 * yl + h1 * ml * dx + h2 * yr + h3 * mr * dx;
    // grid_sample, bilinear, align_corners=True, zero padding; the row coordinate is exact
    const float ix = (ys + 1.0f) * 0.5f * L1;
    const float fx = floorf(ix);
    const int x0 = (int)fx, x1 = x0 + 1;
    const float w1 = ix - fx, w0 = 1.0f - w1;
    const float a0 = (x0 >= 0 && x0 < T) ? row[x0] : 0.f;
    const float a1 = (x1 >= 0 && x1 < T) ? row[x1] : 0.f;
    v = a0 * w0 + a1 * w1;
  } else {
    v = row[t];
  }
  if (t >= t0 && t < t1) v = 0.f;
  if (mI >= f0 && mI < f1) v = 0.f;
  if (mI < lo || mI >= n_mels - hi) v = 0.f;
  out[((long)b * n_mels + mI) * T + t] = v;
}
extern "C" int wft_specaug(const float* in, float* out, const int32_t* params, const int32_t* extremes, int B, int n_mels,
                           int T, void* stream) {
  WFT_CHECK_ARG(in && out && params && in != out, "bad pointers (in and out must differ)");
  WFT_CHECK_ARG(B >= 1 && n_mels >= 1 && T >= 2, "bad shape");
  dim3 grid((T + 255) / 256, n_mels, B);
  hipLaunchKernelGGL(specaug_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, params, extremes, n_mels, T);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// --------------------------------------------------------------------------------- layout for the conv stem
// mel f32 [B, n_mels, T] -> bf16 [B, T+2, c_pad], rows 0 and T+1 zero, channels >= n_mels zero
__global__ __launch_bounds__(256) void mel_tmajor_kernel(const float* mel, unsigned short* out, int n_mels, int T, int c_pad) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, t = t0 + tx;
    tile[cc][tx] = (c < n_mels && t < T) ? mel[((long)b * n_mels + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int tt = ty; tt < 64; tt += 4) {
    const int t = t0 + tt, c = c0 + tx;
    if (t < T && c < c_pad) out[((long)b * (T + 2) + t + 1) * c_pad + c] = f2bf(tile[tx][tt]);
  }
  // halo rows
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < 64; c += 256) {
      const int cg = c0 + c;
      if (cg < c_pad) {
        out[((long)b * (T + 2)) * c_pad + cg] = 0;
        out[((long)b * (T + 2) + T + 1) * c_pad + cg] = 0;
      }
    }
  }
}
extern "C" int wft_mel_to_tmajor_bf16(const float* mel, wft_bf16* out, int B, int n_mels, int T, int c_pad, void* stream) {
  WFT_CHECK_ARG(mel && out, "null pointer");
  WFT_CHECK_ARG(B >= 1 && n_mels >= 1 && T >= 1 && c_pad >= n_mels, "bad shape");
  dim3 grid((T + 63) / 64, (c_pad + 63) / 64, B);
  hipLaunchKernelGGL(mel_tmajor_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, out, n_mels, T, c_pad);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
