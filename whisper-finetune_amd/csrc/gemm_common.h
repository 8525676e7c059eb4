// gemm_common.h — parameter block, tile-order helpers and per-device host helpers shared by the GEMM translation units
// (gemm.hip: 128x128 / 256x256 ping-pong / rank kernels; gemm_nt4w.hip: the one-wave-per-SIMD 256x256 NT kernel).
#pragma once
#include "common.h"

struct GemmP {
  const unsigned short* A; long lda; long sA;
  const unsigned short* B; long ldb; long sB;
  void* C; long ldc; long sC;
  const float* bias;
  const unsigned short* res; long ldr; long sR;
  unsigned short* aux; long ldaux; long sAux;
  float alpha, beta;
  int M, N, K, batch;
  int accumulate;
  int period, valid;
  int res_first;
  float* ws;  // split-K partial tiles [nsplit][P][Q] fp32 (TN, optional)
  float* cs_part;  // NT256: per-(row tile, wave row) column-sum partials [2*tiles_m][N] fp32, or NULL
  int nsplit;  // gemm_tn_rank_kernel: split-K factor (its grid is 1-D)
  int band;  // NT256: tile-order band width in column tiles (WFT_NT256_BAND, default 5)
  int diag;  // WFT_GEMM_DIAG, NT256 A/B switches: 6 skips the staged epilogue (timing only), 7 = general epilogue body everywhere, 8 = no continuous staging
};

// sid -> (row tile, column tile) in column BANDS of 5 tiles, row-major inside a band: the 32 workgroups an XCD
// runs at a time (consecutive sids) then cover a ~6 x 5 patch = 11 operand panels instead of 2 x 20 = 22 for a
// wide N.  Measured before: FETCH_SIZE of the 48000x5120x1280 GEMM was 8x its algorithmic A+B bytes (every XCD
// re-streamed all of B every round).
__device__ __forceinline__ void band_coords(int sid, int tiles_r, int tiles_c, int& tr, int& tc, int W = 5) {
  const int band = sid / (tiles_r * W);
  const int c0 = band * W;
  const int w = (tiles_c - c0) < W ? (tiles_c - c0) : W;
  const int r = sid - band * tiles_r * W;
  tr = r / w;
  tc = c0 + r - tr * w;
}

__device__ __forceinline__ int xcd_remap(int bid, int ntile) {
  const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}


// ---- host side
#define WFT_MAX_DEVICES 64
static inline int wft_cur_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WFT_MAX_DEVICES) dev = 0;
  return dev;
}
// CU count of the CURRENT device (cached per device id: a process may drive several GPUs)
static inline int wft_num_cus() {
  static int n[WFT_MAX_DEVICES] = {0};
  const int dev = wft_cur_device();
  if (n[dev] == 0) {
    hipDeviceProp_t prop;
    int v = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
    n[dev] = v > 0 ? v : 256;
  }
  return n[dev];
}
// hipFuncSetAttribute is per device: remember, per kernel call site, which devices have it
struct DynLdsOnce {
  bool done[WFT_MAX_DEVICES] = {false};
  template <class K>
  bool set(K kfn, int bytes) {  // false (and wft_last_error says why): the caller returns WFT_ERR_LAUNCH instead of launching
    const int dev = wft_cur_device();
    if (!done[dev]) {
      const hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e != hipSuccess) {
        wft_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", bytes, dev, hipGetErrorString(e));
        return false;
      }
      done[dev] = true;
    }
    return true;
  }
};
