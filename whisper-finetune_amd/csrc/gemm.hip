// gemm.hip — bf16 MFMA GEMMs for the Linear / Conv1d-as-GEMM / logits paths.
//
//   wft_gemm_nt_bf16 : C[M,N] = A[M,K] · B[N,K]^T   (forward, backward-data with a
//                      transposed weight shadow)
//   wft_gemm_tn_bf16 : C[P,Q] = A[R,P]^T · B[R,Q]   (weight gradients)
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// 4x4 tiles of v_mfma_f32_16x16x32_bf16), K-step 64, double-buffered LDS filled by
// global_load_lds_dwordx4 (LDS image is lane-linear, so the bank swizzle is applied to
// the per-lane SOURCE address and again on the read — cdna_hip_programming.md rule 21),
// XCD-aware bijective tile remap so that tiles sharing an A row-panel sit on one L2.
// MFMA operands are swapped (D^T = B·A^T) so that each lane ends up with 4 consecutive
// output columns of one row: 8-byte bf16 / 16-byte f32 stores.
#include "common.h"
#include <type_traits>
#ifndef WFT_EPI_PF_CNT
#define WFT_EPI_PF_CNT 4
#endif
#include <stdlib.h>

#include "gemm_common.h"

// ---------------------------------------------------------------------------------- NT
template <int IMM>
__device__ __forceinline__ bf16x8 nt_b128_asm(unsigned lds_byte_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}
// NST = 2: two k-step buffers in 64 KiB of static LDS, two workgroups per CU hide each other's load latency — the form for grids
//   of more than one workgroup per CU.
// NST = 4 (round 6): a ring of four k-step buffers (128 KiB of dynamic LDS, one workgroup per CU) for grids that do NOT fill the
//   chip (a decoder block's Linears at 1 024 rows: 32 workgroups): with a single workgroup per CU the two-buffer form pays one
//   exposed HBM/L2 latency per k-step (measured 0.8-1.5 us per k-step: 47 us for 1 024 x 512 x 2 048, 675 us for the tied-embedding
//   backward-data product 1 024 x 512 x 51 968).  Loads run three k-steps ahead behind counted s_waitcnt vmcnt, plain s_barrier,
//   inline-asm fragment reads (hipcc would drain the LDS-DMA queue in front of its own ds_reads).  Same products in the same order:
//   bit-identical to NST = 2.  With p.nsplit > 1 blockIdx.z is a split of the K range (p.band k-steps each, every split non-empty)
//   and C is the fp32 partial buffer [split][M][ldc] (host: nt_splitk_*; summed in split order by nt_splitk_reduce_kernel).
template <int EPI, bool C_F32, int NST = 2>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_nt_kernel(GemmP p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = p.N >> 7;
  const int tiles_m = (p.M + 127) >> 7;
  const int sid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = sid / tiles_n, tn = sid - tm * tiles_n;
  const int m0 = tm << 7, n0 = tn << 7;
  const int bz = blockIdx.z;
  const bool ksplit = NST > 2 && p.nsplit > 1;
  const int kb = ksplit ? bz * p.band : 0;  // first k-step of this workgroup
  const unsigned short* Ab = ksplit ? p.A + (long)kb * 64 : p.A + (long)bz * p.sA;
  const unsigned short* Bb = ksplit ? p.B + (long)kb * 64 : p.B + (long)bz * p.sB;

  // per-lane source pointers for the 4+4 staging instructions this wave issues per K-tile
  const int lr = lane >> 3, lc = lane & 7;
  const unsigned short* asrc[4];
  const unsigned short* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + lr;
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    asrc[j] = Ab + (long)gm * p.lda + ((lc ^ lr) << 3);
    bsrc[j] = Bb + (long)(n0 + row) * p.ldb + ((lc ^ lr) << 3);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = p.K >> 6;
  const int nk = ksplit ? ((kb + p.band <= nk_all) ? p.band : nk_all - kb) : nk_all;
  const int frow = lane & 15, fg = lane >> 4, sw = lane & 7;
  if constexpr (NST == 2) {
    __shared__ __attribute__((aligned(16))) char smem[65536];  // [buf 2][A 16K | B 16K]
    auto stage = [&](int buf, int kt) {
      char* sa = smem + buf * 32768 + wave * 4096;
      char* sb = sa + 16384;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        glds16(asrc[j] + kt * 64, sa + j * 1024);
        glds16(bsrc[j] + kt * 64, sb + j * 1024);
      }
    };
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
      const char* sa = smem + cur * 32768 + (wm * 64 + frow) * 128;
      const char* sb = smem + cur * 32768 + 16384 + (wn * 64 + frow) * 128;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int coff = ((s * 4 + fg) ^ sw) << 4;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + i * 2048 + coff);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + j * 2048 + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
    extern __shared__ __attribute__((aligned(16))) char dsmem[];  // [slot NST][A 16K | B 16K]
    int ld_slot = 0, ld_k = 0;
    auto stage = [&]() {
      char* sa = dsmem + ld_slot * 32768 + wave * 4096;
      char* sb = sa + 16384;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        glds16(asrc[j] + ld_k * 64, sa + j * 1024);
        glds16(bsrc[j] + ld_k * 64, sb + j * 1024);
      }
      ++ld_k;
      if (++ld_slot == NST) ld_slot = 0;
    };
    // fragment addresses inside a slot: the k half s flips chunk bit 2 (an XOR with the lane's swizzle: one base per half)
    const unsigned lds0 = lds_addr_of(dsmem);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned coff = (unsigned)(((s * 4 + fg) ^ sw) << 4);
      aoff[s] = (unsigned)((wm * 64 + frow) * 128) + coff;
      boff[s] = 16384u + (unsigned)((wn * 64 + frow) * 128) + coff;
    }
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
      if (u < nk) stage();
    int rd_slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // k-step kt has landed when at most the younger k-steps' loads (8 per wave and k-step) are outstanding
      const int ahead = nk - 1 - kt;
      if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 8) : "memory");
      else if (NST == 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading the slot stage() refills now
      if (kt + NST - 1 < nk) stage();
      const unsigned sb = lds0 + rd_slot * 32768;
      bf16x8 af[2][4], bfr[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        af[s][0] = nt_b128_asm<0>(sb + aoff[s]); af[s][1] = nt_b128_asm<2048>(sb + aoff[s]);
        af[s][2] = nt_b128_asm<4096>(sb + aoff[s]); af[s][3] = nt_b128_asm<6144>(sb + aoff[s]);
        bfr[s][0] = nt_b128_asm<0>(sb + boff[s]); bfr[s][1] = nt_b128_asm<2048>(sb + boff[s]);
        bfr[s][2] = nt_b128_asm<4096>(sb + boff[s]); bfr[s][3] = nt_b128_asm<6144>(sb + boff[s]);
      }
      // LDS reads return in order: the first half's 8 fragments are there when 8 reads are still outstanding
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
      if (++rd_slot == NST) rd_slot = 0;
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3] per (i,j)
  const long cb = ksplit ? (long)bz * p.M * p.ldc : (long)bz * p.sC;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + frow;
    if (m >= p.M) continue;
    const bool zero_row = p.period > 0 && (m % p.period) >= p.valid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
      if (p.bias) {
        const f32x4 b4 = *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b4[e];
      }
      if (p.res && p.res_first) {
        const u32x2 r2 = *(const u32x2*)(p.res + (long)bz * p.sR + (long)m * p.ldr + n);
        v[0] += p.beta * bf2f((unsigned short)(r2[0] & 0xffff));
        v[1] += p.beta * bf2f((unsigned short)(r2[0] >> 16));
        v[2] += p.beta * bf2f((unsigned short)(r2[1] & 0xffff));
        v[3] += p.beta * bf2f((unsigned short)(r2[1] >> 16));
      }
      if (EPI == WFT_EPI_GELU) {
        if (p.aux) {
          u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n) = pk;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      } else if (EPI == WFT_EPI_DGELU) {
        const u32x2 a2 = *(const u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n);
        v[0] *= dgelu_f(bf2f((unsigned short)(a2[0] & 0xffff)));
        v[1] *= dgelu_f(bf2f((unsigned short)(a2[0] >> 16)));
        v[2] *= dgelu_f(bf2f((unsigned short)(a2[1] & 0xffff)));
        v[3] *= dgelu_f(bf2f((unsigned short)(a2[1] >> 16)));
      } else if (EPI == WFT_EPI_GELU_GRAD) {
        float dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gelu_both_f(v[e], v[e], dv[e]);
        u32x2 pk = {pack2bf(dv[0], dv[1]), pack2bf(dv[2], dv[3])};
        *(u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n) = pk;
      } else if (EPI == WFT_EPI_MUL_AUX) {
        const u32x2 a2 = *(const u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n);
        v[0] *= bf2f((unsigned short)(a2[0] & 0xffff)); v[1] *= bf2f((unsigned short)(a2[0] >> 16));
        v[2] *= bf2f((unsigned short)(a2[1] & 0xffff)); v[3] *= bf2f((unsigned short)(a2[1] >> 16));
      }
      if (p.res && !p.res_first) {
        const u32x2 r2 = *(const u32x2*)(p.res + (long)bz * p.sR + (long)m * p.ldr + n);
        v[0] += p.beta * bf2f((unsigned short)(r2[0] & 0xffff));
        v[1] += p.beta * bf2f((unsigned short)(r2[0] >> 16));
        v[2] += p.beta * bf2f((unsigned short)(r2[1] & 0xffff));
        v[3] += p.beta * bf2f((unsigned short)(r2[1] >> 16));
      }
      if (zero_row) { v[0] = v[1] = v[2] = v[3] = 0.f; }
      if (C_F32) {
        float* cp = (float*)p.C + cb + (long)m * p.ldc + n;
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if (p.accumulate) {
          const f32x4 old = *(const f32x4*)cp;
          o += old;
        }
        *(f32x4*)cp = o;
      } else {
        unsigned short* cp = (unsigned short*)p.C + cb + (long)m * p.ldc + n;
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)cp = pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- NT 256x256
// Large-shape variant: 256x256 output tile, 512 threads (8 waves as 2(M) x 4(N), 128x64 per wave), one
// workgroup per CU, LDS = ring of four 32-deep k-slabs (4 x {A [256][32], B [256][32]} = 128 KiB).
//
// Measured on the first version (all waves in lockstep, 64-deep tiles): the MFMA pipe was busy 44 % of
// the time; removing the global_load_lds (timing-only build) gave +38 %, i.e. the ~100-cycle issue cost
// of each LDS-DMA instruction was serialised in front of the MFMAs of BOTH waves of a SIMD.  This version
// is a ping-pong: waves 0-3 and 4-7 (SIMD partners) run half a period apart, separated by s_barrier —
//   L-unit: issue 4 global_load_lds (this wave's share of slab u+3) + 12 ds_read_b128 (slab u fragments)
//   C-unit: 32 MFMAs (16x16x32 bf16) on those fragments
// so one partner's loads always sit beside the other partner's MFMAs.  Loads run three slabs (six
// half-periods) ahead behind a COUNTED s_waitcnt vmcnt(8): never drained inside the loop.
__device__ __forceinline__ int nt_g(int row) { return (4 - ((row >> 2) & 3)) & 3; }  // 64-byte-row swizzle

// s_waitcnt for row h of the NT256 epilogue's register ring in its COUNTED body (the asm ties the wait to the registers it
// guards; there is exactly one such statement per half-pass, on no branch).  Vector-memory operations younger than row h's
// load when half-pass h starts — ST stores per half-pass, one ring load per row issued at the end of half-pass h - PF (rows
// 0 .. PF-1 in a prologue); the next tile's LDS-DMA pieces are older than all of them (main-loop tail or before the body):
//   h < PF : rows h+1 .. PF-1, then ST + 1 per half-pass before h
//   h >= PF: half-passes h-PF+1 .. h-1: ST, + 1 while rows remain (j + PF < 16)
template <int ST, int PF>
__device__ __forceinline__ void nt_wait_ring(int h, u32x4& q) {
  int n = 0;
  if (h < PF) n = (PF - 1 - h) + h * (ST + 1);
  else for (int j = h - PF + 1; j < h; ++j) n += ST + (j + PF < 16 ? 1 : 0);
#define WFT_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(q) :: "memory"); break;
  switch (n) {  // h is a compile-time constant after unrolling: one case survives
    WFT_VM_CASE(1) WFT_VM_CASE(2) WFT_VM_CASE(3) WFT_VM_CASE(4) WFT_VM_CASE(5) WFT_VM_CASE(6) WFT_VM_CASE(7) WFT_VM_CASE(8) WFT_VM_CASE(9) WFT_VM_CASE(10) WFT_VM_CASE(11) WFT_VM_CASE(12) WFT_VM_CASE(13) WFT_VM_CASE(14) WFT_VM_CASE(15) WFT_VM_CASE(16) WFT_VM_CASE(17) WFT_VM_CASE(18) WFT_VM_CASE(19) WFT_VM_CASE(20) WFT_VM_CASE(21) WFT_VM_CASE(22) WFT_VM_CASE(23) WFT_VM_CASE(24) WFT_VM_CASE(25) WFT_VM_CASE(26) WFT_VM_CASE(27) WFT_VM_CASE(28) WFT_VM_CASE(29) WFT_VM_CASE(30) WFT_VM_CASE(31) WFT_VM_CASE(32) WFT_VM_CASE(33) WFT_VM_CASE(34) WFT_VM_CASE(35) WFT_VM_CASE(36) WFT_VM_CASE(37) WFT_VM_CASE(38) WFT_VM_CASE(39) WFT_VM_CASE(40)
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(q) :: "memory"); break;
  }
#undef WFT_VM_CASE
}

// WFT_NT_RING slots of 32 KiB (A [256][32] | B [256][32]), LDS-DMA running WFT_NT_RING - 1 slabs ahead of the reads.
// 4 (default): lookahead 3 and a separate 32 KiB staging area for the epilogue.  5 (round 3, built and measured): all 160 KiB are
// ring, lookahead 4 (3.1 us instead of 2.3), slot numbers run on across tiles and the staged epilogue borrows the one slot that
// is free at the seam (the slot of the tile's last slab: the next tile's slabs 0-3 sit in the other four, slab 4 is staged into
// it by the next tile's first L-unit, behind the tile-start barrier).  With operands streamed from HBM (a GEMM run back to back
// on 261 MB activations) the deeper lookahead is worth +3-6 % — in-kernel stamps had shown tiles that open a fresh A panel
// 25 % slower than the others; inside the training step the operands were written just before and come from the Infinity
// Cache: 700.1 vs 701.7 ms per step, no gain (profiles/r03_nt256_ring5_ab.log, r03_step_ring_ab.log).
#ifndef WFT_NT_RING
#define WFT_NT_RING 4
#endif
template <int EPI, bool C_F32>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(GemmP p) {
  constexpr int NSLOT = WFT_NT_RING, LA = NSLOT - 1;  // ring slots, lookahead in slabs
  extern __shared__ __attribute__((aligned(16))) char dsmem[];  // the ring (+ 32 KiB epilogue staging when NSLOT == 4)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool grp_b = wave >= 4;
  const int tiles_n = p.N >> 8;
  const int tiles_m = (p.M + 255) >> 8;
  const int tiles = tiles_m * tiles_n;
  const int total = tiles * p.batch;

  // staging share of this wave: group A (waves 0-3) loads the A part of every slab, group B the B part;
  // wave-instruction = 16 rows x 64 B; this wave owns rows 64*(wave&3) .. +63 of its part (4 instructions)
  const int rr = lane >> 2, cc = lane & 3;
  // Source address of an LDS-DMA piece = wave-uniform tile base (SGPR pair, advanced per slab on the scalar unit) + a
  // per-lane 32-bit byte offset that is constant for the tile: the `saddr + voffset` form, NO vector instruction per
  // piece.  (In-kernel stamps: the 12 reads + 4 pieces of an L-unit took 560-820 cycles to ISSUE — the partner wave's
  // MFMAs run at s_setprio 1 and starve this wave's address arithmetic on the shared VALU port.)
  const char* sbase;
  unsigned soff[4];
  auto set_src = [&](int t) {  // PERSISTENT: tile t of this workgroup's sequence
    const int bz = t / tiles, sid = xcd_remap(t - bz * tiles, tiles);
    int tm, tn;
    band_coords(sid, tiles_m, tiles_n, tm, tn, p.band);
    const unsigned long long b64 = !grp_b ? (unsigned long long)(p.A + (long)bz * p.sA + (long)(tm << 8) * p.lda)
                                          : (unsigned long long)(p.B + (long)bz * p.sB + (long)(tn << 8) * p.ldb);
    // pin the base to SGPRs (it is wave-uniform by construction; the compiler does not prove it through the tile loop)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b64), hi = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
    sbase = (const char*)(((unsigned long long)hi << 32) | lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row = (wave & 3) * 64 + j * 16 + rr;
      const int chunk = cc ^ nt_g(rr);
      if (!grp_b) {
        const int last = p.M - 1 - (tm << 8);  // clamp to the last valid row of A
        row = row < last ? row : last;
        soff[j] = (unsigned)(row * (int)p.lda + chunk * 8) * 2u;
      } else {
        soff[j] = (unsigned)(row * (int)p.ldb + chunk * 8) * 2u;
      }
    }
  };
  const unsigned stage_dst = __builtin_amdgcn_readfirstlane(lds_addr_of(dsmem) + (grp_b ? 16384 : 0) + (wave & 3) * 4096);
  auto stage = [&](int u, int slot_dst) {  // this wave's 4 KiB of slab u of the tile `sbase` points at -> ring slot slot_dst
    const unsigned dst = stage_dst + slot_dst * 32768;
    const unsigned long long sb = (unsigned long long)sbase + (unsigned long long)u * 64;  // wave-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_saddr(soff[j], sb, dst + j * 1024);
  };

  const int nslab = p.K >> 5;
  const int frow = lane & 15, fg = lane >> 4;
  const int coff = (fg ^ nt_g(frow)) << 4;
  const int a_off = (wm * 128 + frow) * 64 + coff;
  const int b_off = 16384 + (wn * 64 + frow) * 64 + coff;
  int slot = 0;  // ring slot of the slab the next L-unit reads; runs on across tiles (wave-uniform scalar)
  auto slot_add = [&](int s_, int d) { const int x = s_ + d; return x >= NSLOT ? x - NSLOT : x; };
  auto prefetch = [&]() {  // shares of slabs 0 .. LA-1 of the tile `src` points at, into the slots the next tile will read
#pragma unroll
    for (int j = 0; j < LA; ++j)
      if (j < nslab) stage(j, slot_add(slot, j));
  };

  int t = blockIdx.x;
  if (t >= total) return;
  set_src(t);
  prefetch();

  for (; t < total; t += gridDim.x) {
    const int bz = t / tiles, sid = xcd_remap(t - bz * tiles, tiles);
    int tm, tn;
    band_coords(sid, tiles_m, tiles_n, tm, tn, p.band);
    const int m0 = tm << 8, n0 = tn << 8;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[8], bq[4];

    // slab 0 complete (its 4 glds are older than everything issued since: epilogue stores, slabs 1 .. LA-1)
    if (nslab >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (LA - 1)) : "memory");
    else if (nslab == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nslab == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp_b) __builtin_amdgcn_s_barrier();  // group B runs half a period behind group A

    // CONTINUOUS staging: with another tile to come and nslab % 4 == 0 (slab j of the next tile then belongs in the slot
    // slab nslab - 4 + j just left), the last three L-units stage the NEXT tile's slabs 0-2 instead of nothing: the 12
    // LDS-DMA pieces are issued beside the partner group's MFMAs like every other slab, not in the epilogue where both
    // groups pay their issue cost with nothing to hide it (stamps: 1-2 us per tile), and the waits never drain.
    const bool more = t + (int)gridDim.x < total;
    const bool cont = more && (NSLOT == 5 || (nslab & 3) == 0) && nslab >= 8 && p.diag != 8;  // (4 slots: slot = u & 3 needs nslab % 4 == 0)
    for (int u = 0; u < nslab; ++u) {
      // ---------------- L-unit (fragment reads first: their latency hides behind the LDS-DMA issue)
      {
        const char* sl = dsmem + slot * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = *(const bf16x8*)(sl + b_off + j * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(sl + a_off + i * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int sdst = slot == 0 ? NSLOT - 1 : slot - 1;  // slot (u + LA) mod NSLOT: slab u - 1 has just left it
        if (u + LA < nslab) {
          stage(u + LA, sdst);
        } else if (cont) {
          if (u + LA == nslab) set_src(t + gridDim.x);  // this tile's source addresses are not needed any more
          stage(u + LA - nslab, sdst);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // slab u+1's share (issued LA L-units ago) must have landed before the partner group reads it
      const int ahead = nslab - 1 - u;
      if (ahead >= LA || cont) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (LA - 1)) : "memory");
      else if (ahead == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- C-unit
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();  // group A idles through group B's last C-unit
    const int free_slot = slot == 0 ? NSLOT - 1 : slot - 1;  // the tile's last slab has left it; nothing is staged into it before the next tile's first L-unit

    // the ring is free: put the next tile's first three slabs in flight, then write this tile out
    const bool staged = !C_F32 && p.diag != 6;
    constexpr bool RD_AUX = (EPI == WFT_EPI_DGELU || EPI == WFT_EPI_MUL_AUX);
    constexpr bool PF_RES = (EPI == WFT_EPI_NONE || EPI == WFT_EPI_GELU);  // the others (no residual in practice) read it in place: registers
    // COUNTED epilogue body: every vector-memory instruction it issues is known (all 128 rows of the wave valid -> every
    // lane active in every half-pass; one ring load per row; EPI_ST stores per half-pass)
    const bool counted = staged && EPI != WFT_EPI_GELU && m0 + wm * 128 + 128 <= p.M && (PF_RES || !p.res) && p.diag != 7;
    if (more && !cont) {  // (older than everything the epilogue issues: outside its counts)
      set_src(t + gridDim.x);
      prefetch();
    }

    const long cb = (long)bz * p.sC;
    if (staged) {
      // ---- epilogue through the 32 KiB of LDS above the ring (4 KiB per wave, one 16-row m-tile per pass,
      // XOR-swizzled 16-byte chunks): every global access below is 16 bytes per lane, 8 lanes = one full
      // 128-byte line (bias / residual / aux / C) instead of 8-byte pieces of 16 different lines.
      //
      // Residual / aux rows are fetched EPI_PF half-passes ahead of their use into a ring of registers.  In the general
      // body (CNT = false) hipcc places the waits, and with row masks and `if (p.res)` around the loads it falls back to
      // vmcnt(0) in front of every use: each of the 16 half-passes drains its own store and the load issued just before
      // it (in-kernel stamps: 18 us per tile with a residual / aux operand from HBM, 6.5 us store-only, 8.5 us with the
      // operand served from L2; main loop 30 - 120 us).  The COUNTED body issues the ring loads as inline asm and waits
      // with hand-counted s_waitcnt vmcnt(N): the counter retires in issue order (loads, stores, LDS-DMA alike), so "at
      // most N younger operations outstanding" is exact when every operation of the body is known.  The two bodies are
      // separate copies of the code: a register that an asm load is still filling must never be copied, and a wait that
      // exists on one side of a branch only makes hipcc copy the ring at the join.
      char* lds = dsmem + (NSLOT == 4 ? 131072 : free_slot * 32768) + wave * 4096;
      const int er = lane >> 3, ec = (lane & 7) * 8;  // row within an 8-row group, first of this lane's 8 columns
      const int ncol = n0 + wn * 64 + ec;
      auto body = [&](auto cnt_c) {
      constexpr bool CNT = decltype(cnt_c)::value;
      constexpr int EPI_PF = CNT ? WFT_EPI_PF_CNT : 4;  // general body: 6 and 8 spill beside the 128 accumulator registers
      constexpr int EPI_ST = (EPI == WFT_EPI_GELU || EPI == WFT_EPI_GELU_GRAD) ? 2 : 1;  // stores per half-pass (GELU: with aux)
      float bias8[8], cs8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; cs8[e] = 0.f; }
      if (p.bias) {
        const f32x4 b0 = *(const f32x4*)(p.bias + ncol), b1 = *(const f32x4*)(p.bias + ncol + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
      }
      // (the counted body of an epilogue that reads the residual in place is only entered without a residual)
      const bool has_res = (CNT && !PF_RES) ? false : (p.res != nullptr);
      const bool ring = RD_AUX || (PF_RES && has_res);
      // row er of this wave's block; later rows = + h * 8 * ld, a wave-uniform step (no 64-bit multiply per access)
      const long row0 = (long)(m0 + wm * 128 + er);
      unsigned short* const c_row0 = (unsigned short*)p.C + cb + row0 * p.ldc + ncol;
      unsigned short* const aux_row0 = p.aux ? p.aux + (long)bz * p.sAux + row0 * p.ldaux + ncol : nullptr;
      const unsigned short* const res_row0 = p.res ? p.res + (long)bz * p.sR + row0 * p.ldr + ncol : nullptr;
      u32x4 auxq[EPI_PF], resq[EPI_PF];
      auto fetch_row = [&](int h, int slot) {
        const int m = m0 + wm * 128 + (h >> 1) * 16 + (h & 1) * 8 + er;
        if (CNT) {  // every row valid
          if (RD_AUX) {
            const unsigned short* src = aux_row0 + (long)(h * 8) * p.ldaux;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(auxq[slot]) : "v"(src) : "memory");
          } else if (PF_RES && has_res) {
            const unsigned short* src = res_row0 + (long)(h * 8) * p.ldr;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(resq[slot]) : "v"(src) : "memory");
          }
        } else if (m < p.M) {
          if (RD_AUX) auxq[slot] = *(const u32x4*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + ncol);
          if (PF_RES && has_res) resq[slot] = *(const u32x4*)(p.res + (long)bz * p.sR + (long)m * p.ldr + ncol);
        }
      };
#pragma unroll
      for (int h = 0; h < EPI_PF; ++h) fetch_row(h, h);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          *(f32x4*)(lds + frow * 256 + (((jj * 4 + fg) ^ frow) << 4)) = acc[i][jj];
        // A wave's LDS instructions execute in order: the reads below see these writes, and the next pass's writes cannot
        // overtake the reads, so the COUNTED body needs no wait between them — it issues the four reads of both half-passes
        // at once and lets hipcc place one counted lgkmcnt wait in front of their first use (one LDS round trip per pass
        // instead of three; two waves per SIMD cannot hide them).  The general body keeps the explicit fences.
        if (!CNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 xa[2], xb[2];
        if (CNT) {
#pragma unroll
          for (int g8 = 0; g8 < 2; ++g8) {
            const int lr = g8 * 8 + er, ch = (lane & 7) * 2;
            xa[g8] = *(const f32x4*)(lds + lr * 256 + ((ch ^ lr) << 4));
            xb[g8] = *(const f32x4*)(lds + lr * 256 + (((ch + 1) ^ lr) << 4));
          }
        }
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
          const int h = i * 2 + g8;
          const int lr = g8 * 8 + er;
          const int m = m0 + wm * 128 + i * 16 + lr;
          const int ch = (lane & 7) * 2;
          const int slot = h % EPI_PF;
          const f32x4 x0 = CNT ? xa[g8] : *(const f32x4*)(lds + lr * 256 + ((ch ^ lr) << 4));
          const f32x4 x1 = CNT ? xb[g8] : *(const f32x4*)(lds + lr * 256 + (((ch + 1) ^ lr) << 4));
          if (CNT && ring) nt_wait_ring<EPI_ST, EPI_PF>(h, RD_AUX ? auxq[slot] : resq[slot]);
          if (CNT || m < p.M) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e] * p.alpha + bias8[e]; v[4 + e] = x1[e] * p.alpha + bias8[4 + e]; }
            const long roff = (long)m;
            u32x4 r4 = resq[slot];
            if (!PF_RES && has_res) r4 = *(const u32x4*)(p.res + (long)bz * p.sR + roff * p.ldr + ncol);
            if (has_res && p.res_first) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += p.beta * bf2f((unsigned short)(r4[e] & 0xffff)); v[2 * e + 1] += p.beta * bf2f((unsigned short)(r4[e] >> 16)); }
            }
            if (EPI == WFT_EPI_GELU) {
              if (p.aux) {
                u32x4 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
                *(u32x4*)(p.aux + (long)bz * p.sAux + roff * p.ldaux + ncol) = pk;
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            } else if (EPI == WFT_EPI_DGELU) {
              const u32x4 a4 = auxq[slot];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] *= dgelu_f(bf2f((unsigned short)(a4[e] & 0xffff)));
                v[2 * e + 1] *= dgelu_f(bf2f((unsigned short)(a4[e] >> 16)));
              }
            } else if (EPI == WFT_EPI_GELU_GRAD) {
              float dv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) gelu_both_f(v[e], v[e], dv[e]);
              u32x4 pk = {pack2bf(dv[0], dv[1]), pack2bf(dv[2], dv[3]), pack2bf(dv[4], dv[5]), pack2bf(dv[6], dv[7])};
              if (CNT) *(u32x4*)(aux_row0 + (long)(h * 8) * p.ldaux) = pk;
              else *(u32x4*)(p.aux + (long)bz * p.sAux + roff * p.ldaux + ncol) = pk;
            } else if (EPI == WFT_EPI_MUL_AUX) {
              const u32x4 a4 = auxq[slot];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] *= bf2f((unsigned short)(a4[e] & 0xffff));
                v[2 * e + 1] *= bf2f((unsigned short)(a4[e] >> 16));
              }
            }
            if (has_res && !p.res_first) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += p.beta * bf2f((unsigned short)(r4[e] & 0xffff)); v[2 * e + 1] += p.beta * bf2f((unsigned short)(r4[e] >> 16)); }
            }
            if (p.period > 0 && (m % p.period) >= p.valid) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            u32x4 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            if (CNT) *(u32x4*)(c_row0 + (long)(h * 8) * p.ldc) = pk;
            else *(u32x4*)((unsigned short*)p.C + cb + roff * p.ldc + ncol) = pk;
            if (p.cs_part) {
#pragma unroll
              for (int e = 0; e < 8; ++e) cs8[e] += v[e];
            }
          }
          if (h + EPI_PF < 16) fetch_row(h + EPI_PF, slot);
        }
        if (!CNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (p.cs_part) {  // column sums of this wave's 128 x 64 block: reduce over the 8 row-lanes, lanes 0-7 store 8 columns each
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t2 = cs8[e];
          t2 += __shfl_xor(t2, 8, 64);
          t2 += __shfl_xor(t2, 16, 64);
          t2 += __shfl_xor(t2, 32, 64);
          cs8[e] = t2;
        }
        if (lane < 8) {
          float* dstp = p.cs_part + (long)(tm * 2 + wm) * p.N + ncol;
          *(f32x4*)dstp = f32x4{cs8[0], cs8[1], cs8[2], cs8[3]};
          *(f32x4*)(dstp + 4) = f32x4{cs8[4], cs8[5], cs8[6], cs8[7]};
        }
      }
      };  // body
      if constexpr (EPI == WFT_EPI_GELU) {  // (conv stem / inference only: its counted copy spills)
        body(std::false_type{});
      } else {
        if (counted) body(std::true_type{}); else body(std::false_type{});
      }
      continue;
    }
    // direct epilogue (fp32 C, accumulate): lane holds C[m][n..n+3] per (i, j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + frow;
      if (m >= p.M) continue;
      const bool zero_row = p.period > 0 && (m % p.period) >= p.valid;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fg * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
        if (p.bias) {
          const f32x4 b4 = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b4[e];
        }
        if (p.res && p.res_first) {
          const u32x2 r2 = *(const u32x2*)(p.res + (long)bz * p.sR + (long)m * p.ldr + n);
          v[0] += p.beta * bf2f((unsigned short)(r2[0] & 0xffff)); v[1] += p.beta * bf2f((unsigned short)(r2[0] >> 16));
          v[2] += p.beta * bf2f((unsigned short)(r2[1] & 0xffff)); v[3] += p.beta * bf2f((unsigned short)(r2[1] >> 16));
        }
        if (EPI == WFT_EPI_GELU) {
          if (p.aux) {
            u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *(u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n) = pk;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        } else if (EPI == WFT_EPI_DGELU) {
          const u32x2 a2 = *(const u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n);
          v[0] *= dgelu_f(bf2f((unsigned short)(a2[0] & 0xffff))); v[1] *= dgelu_f(bf2f((unsigned short)(a2[0] >> 16)));
          v[2] *= dgelu_f(bf2f((unsigned short)(a2[1] & 0xffff))); v[3] *= dgelu_f(bf2f((unsigned short)(a2[1] >> 16)));
        } else if (EPI == WFT_EPI_GELU_GRAD) {
          float dv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) gelu_both_f(v[e], v[e], dv[e]);
          u32x2 pk = {pack2bf(dv[0], dv[1]), pack2bf(dv[2], dv[3])};
          *(u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n) = pk;
        } else if (EPI == WFT_EPI_MUL_AUX) {
          const u32x2 a2 = *(const u32x2*)(p.aux + (long)bz * p.sAux + (long)m * p.ldaux + n);
          v[0] *= bf2f((unsigned short)(a2[0] & 0xffff)); v[1] *= bf2f((unsigned short)(a2[0] >> 16));
          v[2] *= bf2f((unsigned short)(a2[1] & 0xffff)); v[3] *= bf2f((unsigned short)(a2[1] >> 16));
        }
        if (p.res && !p.res_first) {
          const u32x2 r2 = *(const u32x2*)(p.res + (long)bz * p.sR + (long)m * p.ldr + n);
          v[0] += p.beta * bf2f((unsigned short)(r2[0] & 0xffff)); v[1] += p.beta * bf2f((unsigned short)(r2[0] >> 16));
          v[2] += p.beta * bf2f((unsigned short)(r2[1] & 0xffff)); v[3] += p.beta * bf2f((unsigned short)(r2[1] >> 16));
        }
        if (zero_row) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        if (C_F32) {
          float* cp = (float*)p.C + cb + (long)m * p.ldc + n;
          f32x4 o = {v[0], v[1], v[2], v[3]};
          if (p.accumulate) o += *(const f32x4*)cp;
          *(f32x4*)cp = o;
        } else {
          unsigned short* cp = (unsigned short*)p.C + cb + (long)m * p.ldc + n;
          u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2*)cp = pk;
        }
      }
    }
  }
}

// ds_read_b64_tr_b16 from inline asm with an immediate offset (contract as lds_read_tr16_asm in common.h)
template <int IMM>
__device__ __forceinline__ s16x4 tn_tr_asm(unsigned lds_byte_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}

// ---------------------------------------------------------------------------------- TN
// C[p][q] = sum_r A[r][p] * B[r][q].  LDS tiles are [64 r][128 cols] (256-byte rows);
// MFMA operands are column reads of those tiles -> ds_read_b64_tr_b16.
__device__ __forceinline__ int tn_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// PB > 0: only the first 16*PB (<= 64) columns of A are non-zero (a rank-r LoRA operand in its 128-wide padded buffer): the
// wave column wp = 1 and the p-blocks >= PB of wp = 0 skip their fragment reads and MFMAs (their part of C is written as zero).
// NST = 2: two reduction-step buffers in 64 KiB of static LDS, two workgroups per CU (grids of more than one workgroup per CU).
// NST = 4 (round 6, PB = 0): a ring of four buffers in 128 KiB of dynamic LDS for grids of at most one workgroup per CU — a decoder
//   block's weight gradients at R = B*S = 1 024 rows are 16-64 tiles of 16 reduction steps, and the two-buffer form pays an exposed
//   load latency per step there (26 us for 512 x 512 x 1 024, whatever the tile count).  Loads run three steps ahead behind counted
//   s_waitcnt vmcnt, plain s_barrier, inline-asm transposed reads.  Same products in the same order: bit-identical per split.
template <bool C_F32, int PB = 0, int NST = 2>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_tn_kernel(GemmP p) {
  static_assert(NST == 2 || PB == 0, "the ring form is the general kernel only");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave >> 1, wp = wave & 1;
  const int P = p.M, Q = p.N, R = p.K;
  const int tiles_q = Q >> 7;
  const int tiles_p = P >> 7;
  const int sid = xcd_remap(blockIdx.x, tiles_p * tiles_q);
  const int tp = sid / tiles_q, tq = sid - tp * tiles_q;
  const int p0 = tp << 7, q0 = tq << 7;

  const int tpb = (R + 63) >> 6;  // reduction tiles per batch item
  const int nsteps_all = tpb * p.batch;
  // split-K: blockIdx.y owns a contiguous range of reduction steps; partial tiles are summed into C
  // with fp32 atomics issued as whole 256-byte rows (MI355X_MICROARCH.md "Global float atomics")
  const int nsplit = gridDim.y;
  const int per = (nsteps_all + nsplit - 1) / nsplit;
  const int s_begin = blockIdx.y * per;
  const int s_end = (s_begin + per) < nsteps_all ? (s_begin + per) : nsteps_all;
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  char* ep_lds;
  if constexpr (NST == 2) {
    __shared__ __attribute__((aligned(16))) char smem[65536];  // [buf 2][A 16K | B 16K]
    ep_lds = smem;
    // staging: instruction i (0..15) covers r rows 4i..4i+3; lane -> (rr = lane>>4, c' = lane&15)
    const int rr = lane >> 4, cp = lane & 15;
    auto stage = [&](int buf, int step) {
      const int b = step / tpb, t = step - b * tpb;
      const unsigned short* Ab = p.A + (long)b * p.sA;
      const unsigned short* Bb = p.B + (long)b * p.sB;
      char* sa = smem + buf * 32768 + wave * 4096;
      char* sb = sa + 16384;
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = wave * 4 + j;
        const int r = i * 4 + rr;
        int gr = t * 64 + r;
        gr = gr < R ? gr : R - 1;
        const int c = cp ^ (tn_f(r) << 1);
        // rank-r operand: only its first 2*PB 16-byte chunks per row are ever read back (the other LDS slots keep stale bytes)
        if (PB == 0 || c < 2 * PB) glds16(Ab + (long)gr * p.lda + p0 + (c << 3), sa + j * 1024);
        glds16(Bb + (long)gr * p.ldb + q0 + (c << 3), sb + j * 1024);
      }
    };
    auto zero_tail = [&](int buf, int step) {
      const int t = step % tpb;
      const int rem = R - t * 64;  // valid rows in this tile
      if (rem >= 64) return false;
      // rows [rem, 64) of both tiles -> 0 ; 16 chunks of 16 B per row per operand
      char* base = smem + buf * 32768;
      const int nchunk = (64 - rem) * 16;
      for (int c = tid; c < nchunk; c += 256) {
        const int off = (rem * 16 + c) * 16;
        *(u32x4*)(base + off) = u32x4{0, 0, 0, 0};
        *(u32x4*)(base + 16384 + off) = u32x4{0, 0, 0, 0};
      }
      return true;
    };

    stage(0, s_begin);
    __syncthreads();
    if (zero_tail(0, s_begin)) __syncthreads();

    const int g = lane >> 4, li = lane & 15;
    const int r_in = (li >> 2);               // row within the 4-row block
    const int fsw = (r_in | ((g & 1) << 2)) << 1;  // tn_f(r) << 1 for r = 32s + 8g + 4t + r_in
    const int colq = wq * 64 + 4 * (li & 3);  // + iq*16
    const int colp = wp * 64 + 4 * (li & 3);  // + jp*16
    constexpr int NPB = PB > 0 ? PB : 4;      // p-blocks this wave multiplies
    const bool idle = PB > 0 && wp == 1;      // wave-uniform
    for (int step = 0; step < nsteps; ++step) {
      const int cur = step & 1;
      if (step + 1 < nsteps) stage(cur ^ 1, s_begin + step + 1);
      const char* sa = smem + cur * 32768;
      const char* sb = sa + 16384;
      if (!idle) {
  #pragma unroll
      for (int s = 0; s < 2; ++s) {
        s16x8 qf[4], pf[NPB];
  #pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int r = 32 * s + 8 * g + 4 * t + r_in;
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cq = colq + i * 16;
            const int aq = r * 256 + (((cq >> 3) ^ fsw) << 4) + ((cq & 7) << 1);
            const s16x4 x = lds_read_tr16(sb + aq);
  #pragma unroll
            for (int e = 0; e < 4; ++e) qf[i][4 * t + e] = x[e];
            if (i < NPB) {
              const int cpp = colp + i * 16;
              const int ap = r * 256 + (((cpp >> 3) ^ fsw) << 4) + ((cpp & 7) << 1);
              const s16x4 y = lds_read_tr16(sa + ap);
  #pragma unroll
              for (int e = 0; e < 4; ++e) pf[i][4 * t + e] = y[e];
            }
          }
        }
  #pragma unroll
        for (int i = 0; i < 4; ++i)
  #pragma unroll
          for (int j = 0; j < NPB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, qf[i]), __builtin_bit_cast(bf16x8, pf[j]), acc[i][j], 0, 0, 0);
      }
      }
      __syncthreads();
      if (step + 1 < nsteps) {
        if (zero_tail(cur ^ 1, s_begin + step + 1)) __syncthreads();
      }
    }

  } else {
    extern __shared__ __attribute__((aligned(16))) char dsmem[];  // [slot NST][A 16K | B 16K]
    ep_lds = dsmem;
    const int rr = lane >> 4, cp = lane & 15;
    int ld_slot = 0, ld_step = s_begin;
    auto stage = [&]() {
      const int b = ld_step / tpb, t = ld_step - b * tpb;
      const unsigned short* Ab = p.A + (long)b * p.sA;
      const unsigned short* Bb = p.B + (long)b * p.sB;
      char* sa = dsmem + ld_slot * 32768 + wave * 4096;
      char* sb = sa + 16384;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 4 + rr;
        int gr = t * 64 + r;
        gr = gr < R ? gr : R - 1;
        const int c = cp ^ (tn_f(r) << 1);
        glds16(Ab + (long)gr * p.lda + p0 + (c << 3), sa + j * 1024);
        glds16(Bb + (long)gr * p.ldb + q0 + (c << 3), sb + j * 1024);
      }
      ++ld_step;
      if (++ld_slot == NST) ld_slot = 0;
    };
    // fragment read offsets inside a slot for (s, t) = (0, 0); (s, t) adds the immediate 8192 s + 1024 t.  The 16-byte chunk of
    // fragment i is (i ^ r_in) in bits 1-2: lane-dependent, one address register per fragment and operand
    const int g = lane >> 4, li = lane & 15;
    const int r_in = li >> 2;
    const int fsw = (r_in | ((g & 1) << 2)) << 1;
    unsigned qoff[4], poff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cq = wq * 64 + 4 * (li & 3) + i * 16, cpp = wp * 64 + 4 * (li & 3) + i * 16;
      const unsigned rowb = (unsigned)(8 * g + r_in) * 256u;
      qoff[i] = 16384u + rowb + (unsigned)((((cq >> 3) ^ fsw) << 4) + ((cq & 7) << 1));
      poff[i] = rowb + (unsigned)((((cpp >> 3) ^ fsw) << 4) + ((cpp & 7) << 1));
    }
    const unsigned lds0 = lds_addr_of(dsmem);
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
      if (u < nsteps) stage();
    int rd_slot = 0;
    for (int step = 0; step < nsteps; ++step) {
      const int ahead = nsteps - 1 - step;
      if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 8) : "memory");
      else if (NST == 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      {  // rows past R of a batch item's last step: zeros (the loads clamped them to row R - 1)
        const int t = (s_begin + step) % tpb;
        const int rem = R - t * 64;
        if (rem < 64) {  // workgroup-uniform
          char* base = dsmem + rd_slot * 32768;
          const int nchunk = (64 - rem) * 16;
          for (int c = tid; c < nchunk; c += 256) {
            const int off = (rem * 16 + c) * 16;
            *(u32x4*)(base + off) = u32x4{0, 0, 0, 0};
            *(u32x4*)(base + 16384 + off) = u32x4{0, 0, 0, 0};
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      if (step + NST - 1 < nsteps) stage();
      const unsigned sb = lds0 + rd_slot * 32768;
      unsigned qa[4], pa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { qa[i] = sb + qoff[i]; pa[i] = sb + poff[i]; }
      s16x4 qf[2][2][4], pf[2][2][4];  // [s][t][fragment]
#pragma unroll
      for (int i = 0; i < 4; ++i) { qf[0][0][i] = tn_tr_asm<0>(qa[i]); qf[0][1][i] = tn_tr_asm<1024>(qa[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { pf[0][0][i] = tn_tr_asm<0>(pa[i]); pf[0][1][i] = tn_tr_asm<1024>(pa[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { qf[1][0][i] = tn_tr_asm<8192>(qa[i]); qf[1][1][i] = tn_tr_asm<8192 + 1024>(qa[i]); }
      // LDS reads return in order (the counter holds 15 at most): the first half's 16 fragments are there when 8 are outstanding
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { pf[1][0][i] = tn_tr_asm<8192>(pa[i]); pf[1][1][i] = tn_tr_asm<8192 + 1024>(pa[i]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 == 1) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        s16x8 q8[4], p8[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            q8[i][e] = qf[s2][0][i][e]; q8[i][4 + e] = qf[s2][1][i][e];
            p8[i][e] = pf[s2][0][i][e]; p8[i][4 + e] = pf[s2][1][i][e];
          }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q8[i]), __builtin_bit_cast(bf16x8, p8[j]), acc[i][j], 0, 0, 0);
      }
      if (++rd_slot == NST) rd_slot = 0;
    }
    __syncthreads();  // (the epilogue stages through LDS other waves may still be reading)
  }
  const int g = lane >> 4, li = lane & 15;
  constexpr int NPB = PB > 0 ? PB : 4;  // p-blocks this wave multiplied
  if (nsplit > 1) {
    // stage the wave's 64(p) x 64(q) fp32 tile through LDS (two halves of 32 p-rows, row pitch 68
    // floats) so that every atomic wave-instruction adds one contiguous 256-byte row of C
    float* lds = (float*)(ep_lds + wave * 16384);
    if (PB > 0 && p.ws) {
      // rank-r operand: the workspace holds only the 16*PB valid rows of every split, ws[split][16 PB][Q] (P == 128, p0 == 0)
      if (wp == 0) {
        float* wb = p.ws + (long)blockIdx.y * (16 * NPB) * Q + q0 + wq * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < NPB; ++jj) {
#pragma unroll
          for (int i = 0; i < 4; ++i) *(f32x4*)(lds + li * 68 + i * 16 + 4 * g) = acc[i][jj] * p.alpha;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 8
          for (int r = 0; r < 16; ++r) wb[(long)(jj * 16 + r) * Q] = lds[r * 68 + lane];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      return;
    }
    // with a workspace (the default: wft_gemm_tn_workspace_bytes) the partial tile of split blockIdx.y is STORED to
    // ws[split][P][Q] and tn_splitk_reduce_kernel adds the splits in index order: bitwise reproducible.  Without one the
    // partial tiles are added into C with fp32 atomics (order, hence rounding, varies run to run).
    float* cbase = p.ws ? p.ws + ((long)blockIdx.y * P + p0 + wp * 64) * Q + q0 + wq * 64 + lane
                        : (float*)p.C + (long)(p0 + wp * 64) * p.ldc + q0 + wq * 64 + lane;
    const long cld = p.ws ? (long)Q : p.ldc;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *(f32x4*)(lds + (jj * 16 + li) * 68 + i * 16 + 4 * g) = acc[i][half * 2 + jj] * p.alpha;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (p.ws) {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) cbase[(long)(half * 32 + r) * cld] = lds[r * 68 + lane];
      } else {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) atomicAdd(cbase + (long)(half * 32 + r) * cld, lds[r * 68 + lane]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }

  // epilogue: D[q][p]: col (lane&15) = p index, rows 4*(lane>>4)+e = q index
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = p0 + wp * 64 + j * 16 + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = q0 + wq * 64 + i * 16 + g * 4;
      f32x4 o = acc[i][j] * p.alpha;
      if (C_F32) {
        float* cptr = (float*)p.C + (long)pp * p.ldc + qq;
        if (p.accumulate) o += *(const f32x4*)cptr;
        *(f32x4*)cptr = o;
      } else {
        unsigned short* cptr = (unsigned short*)p.C + (long)pp * p.ldc + qq;
        u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *(u32x2*)cptr = pk;
      }
    }
  }
}

template <int V>
struct RankIntC { static constexpr int value = V; };
template <int N, class F>
__device__ __forceinline__ void static_for_rank(F&& f) {
  if constexpr (N > 0) {
    static_for_rank<N - 1>(f);
    f(RankIntC<N - 1>{});
  }
}
// ---------------------------------------------------------------------------------- NT, rank-r B operand
// The other two LoRA adapter products, u = x (s A*mask)^T and du = dy (s B): C[M, 16 PB] = A[M, K] B[16 PB, K]^T with B a rank-r
// operand in the first rows of a 128-row zero-padded buffer.  HBM-bound on A (the activation, read once): the 128-tile
// kernel spends half of its loads in flight on B's zero rows and drains its one-deep prefetch at every __syncthreads; here a
// workgroup streams 128 rows of A through a ring of NST stages {A [128][64 k] 16 KB, B [16 PB][64 k] 2 PB KB} (counted vmcnt,
// plain s_barrier, inline-asm fragment reads) and writes only the 16 PB data columns of the 128-wide C buffer — its consumer,
// gemm_tn_rank_kernel, reads no others.  Same products in the same order as gemm_nt_kernel: bit-identical in those columns.
template <int IMM>
__device__ __forceinline__ bf16x8 lds_b128_asm(unsigned lds_byte_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}
template <int PB>
__global__ __launch_bounds__(256, 2) void gemm_nt_rank_kernel(GemmP pa, GemmP pb, int na) {
  // TWO independent products in one launch (round 3: u = x (sA*m)^T and du = dy (sB) of one adapted Linear group — each alone is
  // 375 workgroups at 32 clips, 1.46 rounds of the chip): workgroups >= na work on the second parameter set.  A single product
  // passes na = gridDim.x.
  const bool second = (int)blockIdx.x >= na;  // workgroup-uniform
  const GemmP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0);
  constexpr int NST = PB <= 2 ? 4 : 3;
  constexpr int BBYTES = 2048 * PB;
  constexpr int SBYTES = 16384 + BBYTES;  // stage = A part, then B part
  constexpr int NBI = (2 * PB + 3) / 4;   // B staging instructions per wave and stage (surplus ones repeat a piece)
  constexpr int LPS = 4 + NBI;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = bid << 7;
  const int lr = lane >> 3, lc = lane & 7;
  // per-lane source pointers of the staging instructions (row of the piece, swizzled 16-byte chunk), advanced by 64 k per step
  const unsigned short* asrc[4];
  const unsigned short* bsrc[NBI];
  int bpiece[NBI];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int gm = m0 + (wave * 4 + j) * 8 + lr;
    gm = gm < p.M ? gm : p.M - 1;
    asrc[j] = p.A + (long)gm * p.lda + ((lc ^ lr) << 3);
  }
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    bpiece[k] = (wave + 4 * k) % (2 * PB);
    bsrc[k] = p.B + (long)(bpiece[k] * 8 + lr) * p.ldb + ((lc ^ lr) << 3);
  }
  int ld_slot = 0, ld_k = 0;
  auto stage = [&]() {
    char* sbase = dsmem + ld_slot * SBYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(asrc[j] + ld_k * 64, sbase + (wave * 4 + j) * 1024);
#pragma unroll
    for (int k = 0; k < NBI; ++k) glds16(bsrc[k] + ld_k * 64, sbase + 16384 + bpiece[k] * 1024);
    ++ld_k;
    if (++ld_slot == NST) ld_slot = 0;
  };

  f32x4 acc[2][PB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fg = lane >> 4, sw = lane & 7;
  const unsigned lds0 = lds_addr_of(dsmem);
  // fragment addresses inside a stage for the k half s = 0; s = 1 flips chunk bit 2 (an XOR, so not an immediate: two bases)
  unsigned aoff[2][2], boff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const unsigned coff = (unsigned)(((s2 * 4 + fg) ^ sw) << 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[s2][i] = (unsigned)((wave * 32 + i * 16 + frow) * 128) + coff;
    boff[s2] = 16384u + (unsigned)(frow * 128) + coff;
  }

  const int nk = p.K >> 6;
#pragma unroll
  for (int u = 0; u < NST - 1; ++u)
    if (u < nk) stage();
  int rd_slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;
    if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
    else if (NST == 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) stage();
    const unsigned sb = lds0 + rd_slot * SBYTES;
    bf16x8 af[2][2], bfr[2][PB];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[s2][i] = lds_b128_asm<0>(sb + aoff[s2][i]);
      static_for_rank<PB>([&](auto jt) {
        constexpr int j = decltype(jt)::value;
        bfr[s2][j] = lds_b128_asm<j * 2048>(sb + boff[s2]);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
    if (++rd_slot == NST) rd_slot = 0;
  }

  // lane holds C[m = 16 i + frow][n = 16 j + 4 fg + e] of the wave's 32 rows
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wave * 32 + i * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const f32x4 v = acc[i][j] * p.alpha;
      const u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *(u32x2*)((unsigned short*)p.C + (long)m * p.ldc + j * 16 + fg * 4) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------- TN, rank-r A operand
// The two LoRA adapter gradients dA = du^T x and dB^T = u^T dy: A is a rank-r operand (du / u, [R, lda]) whose data sits in the
// first 16*PB columns of a 128-wide zero-padded buffer, B the [R, Q] activation stream.  HBM-bound on B (2 bytes per element
// read once; the MFMA work is r/128 of a square tile's), so the kernel is built around the load stream instead of the tile:
//   * ring of NST (4; 3 for PB > 2) stages {B [64 r][128 q] 16 KB, A [64 r][16 PB] 2 PB KB compact}: two workgroups per CU
//     keep 2 x (NST - 1) x 18 KB in flight; LDS-DMA waited for with counted vmcnt and a plain s_barrier per step (the
//     128-tile kernel's __syncthreads is a fence: it drains the prefetch it has just issued)
//   * only the valid rows of a split's partial tile go to the workspace: ws[split][16 PB][Q] (tn_splitk_reduce_kernel adds
//     the splits in index order and writes the padding rows of C as zero)
//   * 1-D grid, split-major through xcd_remap: the q-tiles of one split run on one XCD and share its A rows in that L2.
// All four waves multiply: wave w owns q columns [32 w, 32 w + 32) of the tile and all PB p-blocks.
template <int PB>
__global__ __launch_bounds__(256, 2) void gemm_tn_rank_kernel(GemmP pa, GemmP pb, int na) {
  // two products in one launch (dA = du^T x and dB^T = u^T dy of one adapted group): workgroups >= na take the second set
  const bool second = (int)blockIdx.x >= na;
  const GemmP& p = second ? pb : pa;
  const int bid = (int)blockIdx.x - (second ? na : 0);
  constexpr int NST = PB <= 2 ? 4 : 3;
  constexpr int APITCH = 32 * PB;        // bytes per A row in LDS
  constexpr int ABYTES = 64 * APITCH;    // a stage's A part
  constexpr int SBYTES = 16384 + ABYTES; // stage = B part, then A part
  constexpr int NAI = (2 * PB + 3) / 4;  // A staging instructions per wave and stage (2 PB needed; surplus ones repeat a piece)
  constexpr int LPS = 4 + NAI;           // LDS-DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Q = p.N, R = p.K;
  const int tiles_q = Q >> 7;
  const int nsplit = p.nsplit;
  const int sid = xcd_remap(bid, tiles_q * nsplit);
  const int split = sid / tiles_q, tq = sid - split * tiles_q;
  const int q0 = tq << 7;

  const int tpb = (R + 63) >> 6;  // reduction tiles per batch item
  const int nsteps_all = tpb * p.batch;
  const int per = (nsteps_all + nsplit - 1) / nsplit;
  const int s_begin = split * per;
  const int s_end = (s_begin + per) < nsteps_all ? (s_begin + per) : nsteps_all;
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;  // (the host drops empty splits)

  // staging.  B as in gemm_tn_kernel: instruction i (0..15) covers rows 4i..4i+3, lane -> (rr = lane>>4, position cp = lane&15)
  // holding global chunk cp ^ swizzle(row); A compact row-major: piece ai (1 KB) = chunks 64 ai .. 64 ai + 63 of the
  // [64][2 PB] chunk array.
  const int rr = lane >> 4, cp = lane & 15;
  int ld_b = s_begin / tpb, ld_t = s_begin - ld_b * tpb;
  int ld_slot = 0;
  auto stage = [&]() {
    char* sbase = dsmem + ld_slot * SBYTES;
    const unsigned short* Bb = p.B + (long)ld_b * p.sB + q0;
    const unsigned short* Ab = p.A + (long)ld_b * p.sA;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (wave * 4 + j) * 4 + rr;
      int gr = ld_t * 64 + r;
      gr = gr < R ? gr : R - 1;
      const int c = cp ^ (tn_f(r) << 1);
      glds16(Bb + (long)gr * p.ldb + (c << 3), sbase + wave * 4096 + j * 1024);
    }
#pragma unroll
    for (int k = 0; k < NAI; ++k) {
      const int ai = (wave + 4 * k) % (2 * PB);
      const int id = ai * 64 + lane;
      const int r = id / (2 * PB), c = id - r * (2 * PB);
      int gr = ld_t * 64 + r;
      gr = gr < R ? gr : R - 1;
      glds16(Ab + (long)gr * p.lda + (c << 3), sbase + 16384 + ai * 1024);
    }
    if (++ld_t == tpb) { ld_t = 0; ++ld_b; }
    if (++ld_slot == NST) ld_slot = 0;
  };
  const int rem_last = R - (tpb - 1) * 64;  // valid rows of the last tile of a batch item (64 = full)
  int rd_t = ld_t, rd_slot = 0;

  f32x4 acc[2][PB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, li = lane & 15;
  const int r_in = li >> 2;
  const int fsw = (r_in | ((g & 1) << 2)) << 1;  // tn_f(r) << 1 for r = 32 s + 8 g + 4 t + r_in
  const unsigned lds0 = lds_addr_of(dsmem);
  // per-lane byte offsets inside a stage of the (s = 0, t = 0) reads; t adds 4 rows, s adds 32 rows (immediates)
  unsigned boff[2], aoff[PB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cq = wave * 32 + i * 16 + 4 * (li & 3);
    boff[i] = (unsigned)((8 * g + r_in) * 256 + (((cq >> 3) ^ fsw) << 4) + ((cq & 7) << 1));
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) aoff[j] = (unsigned)(16384 + (8 * g + r_in) * APITCH + (j * 16 + 4 * (li & 3)) * 2);

#pragma unroll
  for (int u = 0; u < NST - 1; ++u)
    if (u < nsteps) stage();

  for (int step = 0; step < nsteps; ++step) {
    const int ahead = nsteps - 1 - step;  // stages issued after this one so far: min(ahead, NST - 2)
    if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");
    else if (NST == 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of this stage have landed; every wave is done reading the previous one
    if (step + NST - 1 < nsteps) stage();
    const unsigned sb = lds0 + rd_slot * SBYTES;
    s16x4 qh[2][2][2], ph[2][PB][2];  // [s][block][t]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const unsigned ad = sb + aoff[j] + s * (32 * APITCH);
        ph[s][j][0] = tn_tr_asm<0>(ad);
        ph[s][j][1] = tn_tr_asm<4 * APITCH>(ad);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned ad = sb + boff[i] + s * 8192;
        qh[s][i][0] = tn_tr_asm<0>(ad);
        qh[s][i][1] = tn_tr_asm<1024>(ad);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool ragged = rd_t == tpb - 1 && rem_last < 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 qf[2], pf[PB];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        s16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = qh[s][i][0][e]; o[4 + e] = qh[s][i][1][e]; }
        qf[i] = __builtin_bit_cast(bf16x8, o);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        s16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = ph[s][j][0][e]; o[4 + e] = ph[s][j][1][e]; }
        pf[j] = __builtin_bit_cast(bf16x8, o);
      }
      if (ragged) {
        // rows >= rem_last of this tile hold clamped duplicates: zero them in ONE operand (element e of the fragment is row
        // 32 s + 8 g + e of the tile)
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (32 * s + 8 * g + e >= rem_last) pf[j][e] = (__bf16)0.0f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], pf[j], acc[i][j], 0, 0, 0);
    }
    if (++rd_t == tpb) rd_t = 0;
    if (++rd_slot == NST) rd_slot = 0;
  }

  // D[q][p]: lane (li, g) holds acc[i][j][e] = C[p = 16 j + li][q = 16 i + 4 g + e] of the wave's 32 columns
  if (p.ws) {  // also with ONE split when the reduce kernel has work of its own (column scale, block-transposed output)
    float* wb = p.ws + (long)split * (16 * PB) * Q + q0 + wave * 32 + 4 * g;
#pragma unroll
    for (int j = 0; j < PB; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) *(f32x4*)(wb + (long)(j * 16 + li) * Q + i * 16) = acc[i][j] * p.alpha;
    return;
  }
  float* cb = (float*)p.C + q0 + wave * 32 + 4 * g;
#pragma unroll
  for (int j = 0; j < PB; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* cptr = cb + (long)(j * 16 + li) * p.ldc + i * 16;
      f32x4 o = acc[i][j] * p.alpha;
      if (p.accumulate) o += *(const f32x4*)cptr;
      *(f32x4*)cptr = o;
    }
  if (!p.accumulate)  // the padding rows of C
    for (int idx = tid; idx < (128 - 16 * PB) * 32; idx += 256) {
      const int row = 16 * PB + (idx >> 5), c4 = (idx & 31) * 4;
      *(f32x4*)((float*)p.C + (long)row * p.ldc + q0 + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---------------------------------------------------------------------------------- TN 256x256
// Weight-gradient GEMM, large-shape variant: 256(p) x 256(q) output tile, 8 waves as 2(q) x 4(p)
// (128 q x 64 p per wave), same ping-pong as gemm_nt256_kernel: ring of four 32-row reduction slabs
// ({A [32 r][256 p], B [32 r][256 q]} = 32 KiB each), waves 0-3 / 4-7 half a period apart, group A stages
// the A part, group B the B part, counted vmcnt(8).  Fragments are transposed LDS reads
// (ds_read_b64_tr_b16) with the pair swizzle of the 128 kernel.  Split-K partials are added to C with
// fp32 atomics issued as contiguous 256-byte half rows staged through LDS.
#ifndef WFT_TN_RING
#define WFT_TN_RING 4  // ring slots of 32 KiB (lookahead = slots - 1); 5 measured +-0.5 % here (profiles/r03_tn_ring5_ab.log): the long reduction loop of one tile per workgroup is not stall-bound the way the NT kernel's fresh operand panels are
#endif
template <bool C_F32>
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(GemmP p) {
  constexpr int NSLOT = WFT_TN_RING, LA = NSLOT - 1;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 2, wp = wave & 3;
  const bool grp_b = wave >= 4;
  const int P = p.M, Q = p.N, R = p.K;
  const int tiles_q = Q >> 8;
  const int tiles_p = P >> 8;
  // 1-D grid over (split, tile) pairs, SPLIT-MAJOR through the XCD map: the hardware deals consecutive workgroup ids round-robin
  // over the 8 XCDs; xcd_remap hands every XCD one contiguous range of pairs, so the ~32 workgroups an XCD runs at a time are
  // tiles of ONE split (or of two neighbours): they walk the same reduction range in step and every A / B slab crosses the
  // fabric once per XCD that needs it, instead of once per XCD for EVERY split (round-2 layout: each split's tiles spread over
  // all 8 XCDs — PMC: 2.27 GB fetched per launch, 2.5x the operands, at 4.1 TB/s)
  const int ntile = tiles_p * tiles_q;
  const int nsplit = p.nsplit < 0 ? -p.nsplit : p.nsplit;
  const bool old_map = p.nsplit < 0;  // round 2's placement (A/B switch)
  const int wsid = old_map ? (int)blockIdx.x : xcd_remap(blockIdx.x, ntile * nsplit);
  const int split = wsid / ntile;
  const int sid = old_map ? xcd_remap(wsid - split * ntile, ntile) : wsid - split * ntile;
  int tp, tq;
  band_coords(sid, tiles_p, tiles_q, tp, tq);
  const int p0 = tp << 8, q0 = tq << 8;

  const int spb = (R + 31) >> 5;  // 32-row slabs per batch item
  const int nslab_all = spb * p.batch;
  const int per = (nslab_all + nsplit - 1) / nsplit;
  const int s_begin = split * per;
  const int s_end = (s_begin + per) < nslab_all ? (s_begin + per) : nslab_all;
  const int nslab = s_end - s_begin;
  if (nslab <= 0) return;

  // staging share: one wave-instruction = 2 rows of 512 B; a part = 16 instructions; wave (w & 3) of the
  // group owns instructions 4(w&3) .. +3 = rows 8(w&3) .. +7 of its part
  const int rr = lane >> 5, cp = lane & 31;
  const unsigned short* const gbase = grp_b ? p.B : p.A;
  const long gld = grp_b ? p.ldb : p.lda;
  const long gbs = grp_b ? p.sB : p.sA;
  const int col0 = grp_b ? q0 : p0;
  char* const stage_dst = dsmem + (grp_b ? 16384 : 0) + (wave & 3) * 4096;
  const unsigned stage_dst_s = __builtin_amdgcn_readfirstlane(lds_addr_of(stage_dst));
  // full slabs: scalar base + constant per-lane byte offset (saddr form, no vector instruction per piece; see the NT kernel)
  unsigned soff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave & 3) * 8 + j * 2 + rr;
    soff[j] = (unsigned)(r * (int)gld + ((cp ^ (tn_f(r) << 1)) << 3)) * 2u;
  }
  // (batch item, slab-in-item) of the next slab to stage / to read, advanced incrementally (no division in the loop)
  int ld_b = s_begin / spb, ld_t = s_begin - ld_b * spb;
  int rd_t = ld_t;
  int ld_slot = 0, rd_slot = 0;  // ring slots of the next slab to stage / to read (stage() is called in slab order)
  auto stage = [&](int u) {  // local slab index u -> ring slot u mod NSLOT; rows past R are clamped (masked at read time)
    const unsigned short* base = gbase + (long)ld_b * gbs + col0;
    if (ld_t * 32 + 32 <= R) {
      const unsigned long long b64 = (unsigned long long)(base + (long)ld_t * 32 * gld);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b64), hi = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
      const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
      const unsigned dsts = stage_dst_s + ld_slot * 32768;
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16_saddr(soff[j], sb, dsts + j * 1024);
    } else {
      char* dst = stage_dst + ld_slot * 32768;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (wave & 3) * 8 + j * 2 + rr;
        int gr = ld_t * 32 + r;
        gr = gr < R ? gr : R - 1;
        const int c = cp ^ (tn_f(r) << 1);
        glds16(base + (long)gr * gld + (c << 3), dst + j * 1024);
      }
    }
    if (++ld_t == spb) { ld_t = 0; ++ld_b; }
    ld_slot = ld_slot + 1 == NSLOT ? 0 : ld_slot + 1;
  };
  const int rem_last = R - (spb - 1) * 32;  // valid rows of the last slab of a batch item (32 = full)

  f32x4 acc[8][4];  // [q tile][p tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, li = lane & 15;
  const int r_in = li >> 2;
  const int fsw = (r_in | ((g & 1) << 2)) << 1;
  const int colq = wq * 128 + 4 * (li & 3);  // + i*16
  const int colp = wp * 64 + 4 * (li & 3);   // + j*16
  const unsigned lds0 = lds_addr_of(dsmem);
  // per-lane byte offsets (within a slab part) of the two transposed reads of a fragment at column `col`
  auto frag_off = [&](int col, int t) -> unsigned {
    const int r = 8 * g + 4 * t + r_in;
    return (unsigned)(r * 512 + (((col >> 3) ^ fsw) << 4) + ((col & 7) << 1));
  };
  s16x4 qh[8][2], ph[4][2];  // raw halves of the fragments (inline-asm reads: waited for by hand below)

#pragma unroll
  for (int j = 0; j < LA; ++j)
    if (j < nslab) stage(j);
  if (nslab >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (LA - 1)) : "memory");
  else if (nslab == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nslab == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp_b) __builtin_amdgcn_s_barrier();

  bf16x8 qf[8], pf[4];
  for (int u = 0; u < nslab; ++u) {
    // ---------------- L-unit
    if (u + LA < nslab) stage(u + LA);
    {
      const unsigned sa = lds0 + rd_slot * 32768;
      const unsigned sb = sa + 16384;
      // the second 4-row group of a fragment is +4 rows = +2048 B: in the instruction's immediate, not a second address
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned ad = sa + frag_off(colp + j * 16, 0);
        ph[j][0] = tn_tr_asm<0>(ad);
        ph[j][1] = tn_tr_asm<2048>(ad);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned ad = sb + frag_off(colq + i * 16, 0);
        qh[i][0] = tn_tr_asm<0>(ad);
        qh[i][1] = tn_tr_asm<2048>(ad);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = ph[j][0][e]; o[4 + e] = ph[j][1][e]; }
      pf[j] = __builtin_bit_cast(bf16x8, o);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = qh[i][0][e]; o[4 + e] = qh[i][1][e]; }
      qf[i] = __builtin_bit_cast(bf16x8, o);
    }
    if (rd_t == spb - 1 && rem_last < 32) {
      // ragged end of the reduction: rows >= rem_last of this slab hold clamped duplicates; zero them in
      // ONE operand (element 4t+e of the fragment is row 8g + 4t + e)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * g + e >= rem_last) pf[j][e] = (__bf16)0.0f;
    }
    if (++rd_t == spb) rd_t = 0;
    const int ahead = nslab - 1 - u;
    if (ahead >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (LA - 1)) : "memory");
    else if (ahead == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- C-unit
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], pf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    rd_slot = rd_slot + 1 == NSLOT ? 0 : rd_slot + 1;
  }
  if (!grp_b) __builtin_amdgcn_s_barrier();

  // D[q][p]: lane (li, g) holds acc[i][j][e] = C[p = j*16 + li][q = i*16 + 4g + e] of the wave tile
  if (nsplit > 1 || p.ws) {  // (a workspace with ONE split: segmented output, written by the reduce kernel)
    float* lds = (float*)(dsmem + wave * 8448);  // [16 p][132] fp32 per pass
    if (p.ws) {
      // deterministic split-K: this split's partial tile goes to the workspace with plain 16-byte stores
      // (rows of 128 fp32 = 512 B per wave: 32 lanes x 16 B), summed later in split order
      float* wbase = p.ws + ((long)split * P + p0 + wp * 64) * Q + q0 + wq * 128;
      const int hr = lane >> 5, c4 = (lane & 31) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4*)(lds + li * 132 + i * 16 + 4 * g) = acc[i][j] * p.alpha;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          *(f32x4*)(wbase + (long)(j * 16 + r + hr) * Q + c4) = *(const f32x4*)(lds + (r + hr) * 132 + c4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
    float* cbase = (float*)p.C + (long)(p0 + wp * 64) * p.ldc + q0 + wq * 128 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f32x4*)(lds + li * 132 + i * 16 + 4 * g) = acc[i][j] * p.alpha;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        float* crow = cbase + (long)(j * 16 + r) * p.ldc;
        atomicAdd(crow, lds[r * 132 + lane]);
        atomicAdd(crow + 64, lds[r * 132 + 64 + lane]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = p0 + wp * 64 + j * 16 + li;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int qq = q0 + wq * 128 + i * 16 + g * 4;
      f32x4 o = acc[i][j] * p.alpha;
      if (C_F32) {
        float* cptr = (float*)p.C + (long)pp * p.ldc + qq;
        if (p.accumulate) o += *(const f32x4*)cptr;
        *(f32x4*)cptr = o;
      } else {
        unsigned short* cptr = (unsigned short*)p.C + (long)pp * p.ldc + qq;
        u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *(u32x2*)cptr = pk;
      }
    }
  }
}

// split-K reduction: C[p][q] (+)= sum_s ws[s][p][q], splits added in index order (reproducible)
// Rows P .. Pz-1 of C (the padding rows of a rank-r operand, absent from the workspace) are written as zero.
// col_scale / blk_n: the LoRA adapter-gradient forms of wft_gemm_args (tn_col_scale, tn_block_n): a per-(row group, column)
// factor, and the transposed block-diagonal output.
struct TnReduceP {
  const float* ws; float* C; long ldc; int P, Q, nsplit, accumulate, Pz; const float* col_scale; int scale_rows, blk_n, blk_r, Pv;
};
__device__ __forceinline__ void tn_splitk_reduce_body(const float* ws, float* C, long ldc, int P, int Q, int nsplit,
                                                      int accumulate, int Pz, const float* col_scale, int scale_rows,
                                                      int blk_n, int blk_r, int Pv, long bid, long nblk) {
  const long nq4 = Q >> 2;
  const long total = (long)(Pz > P ? Pz : P) * nq4;
  for (long i = bid * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const long pp = i / nq4, q4 = (i - pp * nq4) * 4;
    float* cp = C + pp * ldc + q4;
    if (col_scale && blk_n == 0 && pp >= Pv) continue;  // col-scale form: C has p_valid rows, nothing is written below them
    if (pp >= P) {
      if (!accumulate && blk_n == 0) *(f32x4*)cp = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    int pr = 0;
    float* ob = nullptr;
    if (blk_n > 0) {
      const int blk = (int)(q4 / blk_n);  // q4 .. q4+3 lie in one block (blk_n % 4 == 0)
      pr = (int)pp - blk * blk_r;
      if (pr < 0 || pr >= blk_r) continue;  // off-diagonal: not stored
      ob = C + (long)blk * blk_n * blk_r + (q4 - (long)blk * blk_n) * blk_r + pr;
    }
    const bool plain = !col_scale && !ob;
    // plain form: C (+)= sum of the splits, starting from C when accumulating (the order the 256x256 path always had)
    f32x4 s = (plain && accumulate) ? *(const f32x4*)cp : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) s += *(const f32x4*)(ws + ((long)k * P + pp) * Q + q4);
    if (plain) {
      *(f32x4*)cp = s;
      continue;
    }
    // rows >= Pv (p_valid rounded up to 16 leaves up to 15 of them) are sums of zeros and have no scale row
    if (col_scale && pp < Pv) s *= *(const f32x4*)(col_scale + (scale_rows > 0 ? pp / scale_rows : 0) * Q + q4);
    if (ob) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[(long)e * blk_r] = accumulate ? ob[(long)e * blk_r] + s[e] : s[e];
    } else {
      if (accumulate) s += *(const f32x4*)cp;
      *(f32x4*)cp = s;
    }
  }
}

__global__ __launch_bounds__(256) void tn_splitk_reduce_kernel(const float* ws, float* C, long ldc, int P, int Q, int nsplit,
                                                                int accumulate, int Pz, const float* col_scale, int scale_rows,
                                                                int blk_n, int blk_r, int Pv) {
  tn_splitk_reduce_body(ws, C, ldc, P, Q, nsplit, accumulate, Pz, col_scale, scale_rows, blk_n, blk_r, Pv, (long)blockIdx.x, (long)gridDim.x);
}
// the plain form with SEGMENTED output (wft_gemm_args tn_seg_*): row range i of the product goes to its own contiguous [rows][Q]
// tensor — each parameter of a fused Linear group gets its gradient where it lives (a DDP bucket view).  Same order of additions
// as tn_splitk_reduce_body: C first when accumulating, then the splits in index order.
struct TnSegs { float* ptr[4]; int end[4]; };
__global__ __launch_bounds__(256) void tn_splitk_reduce_seg_kernel(const float* ws, TnSegs sg, int P, int Q, int nsplit, int accumulate) {
  const long nq4 = Q >> 2;
  const long total = (long)P * nq4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pp = i / nq4, q4 = (i - pp * nq4) * 4;
    int sgi = 0, start = 0;
    while (sgi < 3 && pp >= sg.end[sgi]) start = sg.end[sgi++];
    float* cp = sg.ptr[sgi] + (pp - start) * Q + q4;
    f32x4 s = accumulate ? *(const f32x4*)cp : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) s += *(const f32x4*)(ws + ((long)k * P + pp) * Q + q4);
    *(f32x4*)cp = s;
  }
}
// both adapter gradients of a group (dA with its column scale, dB in block layout) in ONE launch: workgroups >= g0 take r1
__global__ __launch_bounds__(256) void tn_splitk_reduce_pair_kernel(TnReduceP r0, TnReduceP r1, int g0) {
  const bool second = (int)blockIdx.x >= g0;
  const TnReduceP& r = second ? r1 : r0;
  tn_splitk_reduce_body(r.ws, r.C, r.ldc, r.P, r.Q, r.nsplit, r.accumulate, r.Pz, r.col_scale, r.scale_rows, r.blk_n, r.blk_r, r.Pv,
                        (long)blockIdx.x - (second ? g0 : 0), (long)(second ? (int)gridDim.x - g0 : g0));
}

// ---------------------------------------------------------------------------------- host
static int g_diag = 0;
// dispatch thresholds, measured at M = R = 4096 and 8704 (decoder-sized problems; tests/dev_small_gemm.py): the 256x256 kernels win once
// they can occupy half of the CUs (NT: >= 128 tiles) / have >= 50 output tiles to split (TN).  Tunable: WFT_NT256_MIN_TILES, WFT_TN256_MIN_STEPS.
static int g_nt256_min_tiles = 128, g_tn256_min_steps = 64, g_tn256_min_out_tiles = 50;
static bool g_nt256_persistent = true;
static int g_nt256_band = 5;
// which 256x256 NT kernel: 0 = the one-wave-per-SIMD kernel where it applies (gemm_nt4w.hip), 1 = always the ping-pong kernel.
// WFT_NT_VARIANT=pp|4w at load time (timing builds); per call: wft_gemm_args.variant.
static int g_nt_variant = 0;
bool wft_nt4w_eligible(const wft_gemm_args* a);
// the one-wave-per-SIMD weight-gradient kernel (gemm_tn4w.hip): WFT_TN_VARIANT=pp keeps gemm_tn256_kernel
static int g_tn_variant = 0;
bool wft_tn4w_eligible(const wft_gemm_args* a);
void wft_tn4w_plan(const wft_gemm_args* a, int* nsplit_out, int* per_out);
int wft_tn4w_launch(const wft_gemm_args* a, GemmP p, int nsplit, int per, void* stream);

int wft_nt4w_launch(const wft_gemm_args* a, const GemmP& p, bool persistent, void* stream);


static bool g_force_128 = false;  // debugging / A-B switch: WFT_GEMM_FORCE_128=1
static struct EnvInit { EnvInit() { const char* e = wft_dev_getenv("WFT_GEMM_FORCE_128"); g_force_128 = e && e[0] == '1'; const char* d = wft_dev_getenv("WFT_GEMM_DIAG"); g_diag = d ? atoi(d) : 0; const char* t1 = wft_dev_getenv("WFT_NT256_MIN_TILES"); if (t1) g_nt256_min_tiles = atoi(t1); const char* t2 = wft_dev_getenv("WFT_TN256_MIN_STEPS"); if (t2) g_tn256_min_steps = atoi(t2); const char* t3 = wft_dev_getenv("WFT_TN256_MIN_OUT_TILES"); if (t3) g_tn256_min_out_tiles = atoi(t3); const char* pe = getenv("WFT_NT256_PERSISTENT"); if (pe) g_nt256_persistent = pe[0] != '0'; const char* bw = wft_dev_getenv("WFT_NT256_BAND"); if (bw && atoi(bw) > 0) g_nt256_band = atoi(bw); const char* nv = wft_dev_getenv("WFT_NT_VARIANT"); if (nv) g_nt_variant = (nv[0] == 'p') ? 1 : 0; const char* tv = wft_dev_getenv("WFT_TN_VARIANT"); if (tv) g_tn_variant = (tv[0] == 'p') ? 1 : 0; } } g_env_init;



static int fill_params(const wft_gemm_args* a, GemmP& p) {
  p.A = a->A; p.lda = a->lda; p.sA = a->strideA;
  p.B = a->B; p.ldb = a->ldb; p.sB = a->strideB;
  p.C = a->C; p.ldc = a->ldc; p.sC = a->strideC;
  p.bias = a->bias;
  p.res = a->residual; p.ldr = a->ldr; p.sR = a->strideR;
  p.aux = a->aux; p.ldaux = a->ldaux; p.sAux = a->strideAux;
  p.alpha = a->alpha;
  p.beta = a->beta == 0.f ? 1.f : a->beta;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K; p.batch = a->batch;
  p.accumulate = a->accumulate;
  p.period = a->valid_rows_period; p.valid = a->valid_rows;
  p.res_first = a->residual_first;
  p.diag = g_diag;
  p.band = g_nt256_band;
  p.ws = nullptr;
  p.cs_part = nullptr;
  p.nsplit = 1;
  return 0;
}

// out[col] = sum over `nrows` partial rows (fixed order): finishes the fused bias-gradient column sums of gemm_nt256_kernel
__global__ __launch_bounds__(256) void nt_colsum_reduce_kernel(const float* partial, int nrows, int n, float* out) {
  // 64 columns per workgroup as 16 groups of four (16-byte loads: a wave instruction covers four whole 256-byte row segments),
  // 16 row lanes; round 5: the 32-column / 4-byte form streamed its 20 MB at 0.7 TB/s (29.8 us per fc2 backward-data GEMM)
  __shared__ f32x4 red[16][17];
  const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + cg * 4;
  f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (col < n)
    for (int r = ry; r < nrows; r += 16) sacc += *(const f32x4*)(partial + (long)r * n + col);
  red[ry][cg] = sacc;
  __syncthreads();
  if (ry == 0 && col < n) {
    f32x4 t = red[0][cg];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][cg];
    *(f32x4*)(out + col) = t;
  }
}

static bool nt_uses_256(const wft_gemm_args* a);
// C[m][n] = bf16(sum over splits, in split order, of ws[split][m][n]): finishes the split-K form of the 128-tile NT kernel
__global__ __launch_bounds__(256) void nt_splitk_reduce_kernel(const float* ws, int nsplit, int M, int N, unsigned short* C, long ldc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per 4 consecutive columns
  const int n4 = N >> 2;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (long)m * n4) << 2;
  const float* src = ws + (long)m * N + n;
  f32x4 t = *(const f32x4*)src;
  for (int k = 1; k < nsplit; ++k) t += *(const f32x4*)(src + (long)k * M * N);
  const u32x2 pk = {pack2bf(t[0], t[1]), pack2bf(t[2], t[3])};
  *(u32x2*)(C + (long)m * ldc + n) = pk;
}
// 128-tile NT problems whose grid leaves most CUs idle over a deep K (the tied-embedding backward-data product of a short decoder
// batch: 1 024 x 512 x 51 968 = 32 tiles of 812 k-steps): K is split over the idle CUs, fp32 partial tiles go to the caller's
// workspace and are summed in split order (bitwise reproducible).  Plain products only, and not N = 128: those are the rank-r adapter
// products, which stay bit-identical to their p_valid form (gemm_nt_rank_kernel).  Returns the split count (1 = unsplit).
static int nt_splitk_plan(const wft_gemm_args* a, int* per_out) {
  if (a->c_is_f32 || a->batch != 1 || a->epilogue != WFT_EPI_NONE || a->bias || a->residual || a->aux || a->colsum ||
      a->valid_rows_period != 0 || a->p_valid != 0 || a->N % 128 != 0 || a->N == 128 || g_diag == 11 || g_diag == 12)
    return 1;
  const long tiles = ((a->M + 127) / 128) * (a->N / 128), nk = a->K / 64;
  const int ncu = wft_num_cus();
  if (tiles * 2 > ncu || nk < 64) return 1;  // (at least two splits' worth of idle CUs)
  long nsplit = ncu / tiles;
  if (nsplit > nk / 16) nsplit = nk / 16;
  const long per = (nk + nsplit - 1) / nsplit;
  *per_out = (int)per;
  return (int)((nk + per - 1) / per);  // (no empty split)
}
extern "C" int64_t wft_gemm_nt_splitk_workspace_bytes(const wft_gemm_args* a) {
  if (!a || nt_uses_256(a)) return 0;
  int per = 0;
  const int ns = nt_splitk_plan(a, &per);
  return ns > 1 ? (int64_t)ns * a->M * a->N * 4 : 0;
}
static bool nt_uses_256(const wft_gemm_args* a) {
  const bool wide_ok = a->c_is_f32 || (a->ldc % 8 == 0 && (!a->residual || (a->ldr % 8 == 0 && ((uintptr_t)a->residual & 15) == 0)) &&
                                       (!a->aux || (a->ldaux % 8 == 0 && ((uintptr_t)a->aux & 15) == 0)) &&
                                       (!a->bias || ((uintptr_t)a->bias & 15) == 0));
  return !g_force_128 && wide_ok && a->N % 256 == 0 && a->M >= 1024 &&
         ((a->M + 255) / 256) * (a->N / 256) * a->batch >= g_nt256_min_tiles;
}
extern "C" int wft_gemm_nt_variant(const wft_gemm_args* a) {
  if (!a || !nt_uses_256(a)) return 128;
  return (g_nt_variant != 1 && a->variant == 0 && wft_nt4w_eligible(a)) ? 4 : 256;
}

// bytes of the fragment-ordered one-byte gelu' buffer (WFT_EPI_GELU_GRAD8 writes, WFT_EPI_MUL_AUX8 reads) if these arguments are
// served — only gemm_nt4w_kernel carries the two epilogues — else 0: 16 KiB per (256x256 tile, wave) = one byte per tile element
extern "C" int64_t wft_gemm_nt_aux8_bytes(const wft_gemm_args* a) {
  if (!a || (a->epilogue != WFT_EPI_GELU_GRAD8 && a->epilogue != WFT_EPI_MUL_AUX8)) return 0;
  if (!nt_uses_256(a) || g_nt_variant == 1 || a->variant != 0 || !wft_nt4w_eligible(a)) return 0;
  return (int64_t)((a->M + 255) / 256) * (a->N / 256) * 65536;
}

extern "C" int64_t wft_gemm_nt_colsum_workspace_bytes(const wft_gemm_args* a) {
  if (!a || !a->colsum || a->c_is_f32 || a->batch != 1 || !nt_uses_256(a)) return 0;
  return (int64_t)2 * ((a->M + 255) / 256) * a->N * (int64_t)sizeof(float);
}

extern "C" int wft_gemm_nt_bf16(const wft_gemm_args* a, void* stream) {
  WFT_CHECK_ARG(a && a->A && a->B && a->C, "null pointer");
  WFT_CHECK_ARG(a->M >= 1 && a->N >= 128 && a->K >= 64 && a->batch >= 1, "bad shape");
  WFT_CHECK_ARG(a->N % 128 == 0, "N must be a multiple of 128");
  WFT_CHECK_ARG(a->K % 64 == 0, "K must be a multiple of 64");
  WFT_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 4 == 0, "ld alignment");
  WFT_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0,
                "base pointers must be 16-byte aligned");
  WFT_CHECK_ARG(!(a->accumulate && !a->c_is_f32), "accumulate needs an f32 C");
  const bool aux8 = a->epilogue == WFT_EPI_GELU_GRAD8 || a->epilogue == WFT_EPI_MUL_AUX8;
  WFT_CHECK_ARG((a->epilogue != WFT_EPI_DGELU && a->epilogue != WFT_EPI_GELU_GRAD && a->epilogue != WFT_EPI_MUL_AUX && !aux8) || a->aux,
                "DGELU / GELU_GRAD / MUL_AUX epilogues need aux");
  WFT_CHECK_ARG((a->epilogue != WFT_EPI_GELU_GRAD && a->epilogue != WFT_EPI_MUL_AUX && !aux8) || !a->c_is_f32,
                "GELU_GRAD / MUL_AUX epilogues write a bf16 C");
  if (aux8 && wft_gemm_nt_aux8_bytes(a) == 0) {
    wft_set_error("wft_gemm_nt_bf16: the one-byte gelu' epilogues exist on gemm_nt4w_kernel only (ask wft_gemm_nt_aux8_bytes first)");
    return WFT_ERR_UNSUPPORTED;
  }
  WFT_CHECK_ARG(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "dims exceed int32");
  GemmP p;
  fill_params(a, p);
  hipStream_t s = (hipStream_t)stream;
  // big, 256-aligned-N problems go to the 256x256 kernel (one workgroup per CU, 128 KiB LDS)
  const bool big = nt_uses_256(a);
  const bool cs_fused = a->colsum && big && !a->c_is_f32 && a->batch == 1 && a->workspace &&
                        a->workspace_bytes >= wft_gemm_nt_colsum_workspace_bytes(a) &&
                        ((((uintptr_t)a->workspace) | ((uintptr_t)a->colsum)) & 15) == 0;  // (16-byte accesses in the reduce kernel)
  if (a->colsum) {
    WFT_CHECK_ARG(!a->c_is_f32 && a->batch == 1, "colsum needs a bf16 C and batch == 1");
    if (cs_fused) p.cs_part = (float*)a->workspace;
  }
  // launch state is per call (wft_gemm_args.launch_mode / variant); the process-wide start values come from the environment at load
  // time only (WFT_NT256_PERSISTENT; the variant variables in timing builds) and never change afterwards
  const bool persistent = g_nt256_persistent && a->launch_mode != 1;
  if (big && g_nt_variant != 1 && a->variant == 0 && wft_nt4w_eligible(a)) {
    const int rc = wft_nt4w_launch(a, p, persistent, stream);
    if (rc != WFT_OK) return rc;
    if (cs_fused)
      hipLaunchKernelGGL(nt_colsum_reduce_kernel, dim3((unsigned)((a->N + 63) / 64)), dim3(256), 0, s, (const float*)a->workspace,
                         (int)(2 * ((a->M + 255) / 256)), (int)a->N, a->colsum);
    WFT_CHECK_LAUNCH();
    if (a->colsum && !cs_fused) return wft_colsum_bf16((const wft_bf16*)a->C, a->M, a->N, a->ldc, a->colsum, 0, stream);
    return WFT_OK;
  }
  if (big) {
    const long t256 = ((a->M + 255) / 256) * (a->N / 256) * a->batch;
    const int ncu = wft_num_cus();
    // persistent (one workgroup per CU walks the tiles, prefetching across tile seams) unless WFT_NT256_PERSISTENT=0: with
    // collectives running beside the GEMMs (DDP over RCCL) some CUs are busy when the kernel starts, and a static tile
    // walk would leave their share for the end; one workgroup per tile lets the hardware dispatcher balance instead
    dim3 grid((unsigned)((t256 < ncu || !persistent) ? t256 : ncu)), block(512);
#define LAUNCH_256(E, F)                                                                                   \
  do {                                                                                                    \
    auto kfn = gemm_nt256_kernel<E, F>;                                                                   \
    static DynLdsOnce once;                                                                               \
    if (!once.set(kfn, 163840)) return WFT_ERR_LAUNCH;                                                                                \
    hipLaunchKernelGGL(kfn, grid, block, 163840, s, p);                                                   \
  } while (0)
    switch (a->epilogue) {
      case WFT_EPI_NONE: if (a->c_is_f32) LAUNCH_256(WFT_EPI_NONE, true); else LAUNCH_256(WFT_EPI_NONE, false); break;
      case WFT_EPI_GELU: if (a->c_is_f32) LAUNCH_256(WFT_EPI_GELU, true); else LAUNCH_256(WFT_EPI_GELU, false); break;
      case WFT_EPI_DGELU: if (a->c_is_f32) LAUNCH_256(WFT_EPI_DGELU, true); else LAUNCH_256(WFT_EPI_DGELU, false); break;
      case WFT_EPI_GELU_GRAD: LAUNCH_256(WFT_EPI_GELU_GRAD, false); break;
      case WFT_EPI_MUL_AUX: LAUNCH_256(WFT_EPI_MUL_AUX, false); break;
      default: wft_set_error("wft_gemm_nt_bf16: unknown epilogue %d", a->epilogue); return WFT_ERR_ARG;
    }
#undef LAUNCH_256
    if (cs_fused)
      hipLaunchKernelGGL(nt_colsum_reduce_kernel, dim3((unsigned)((a->N + 63) / 64)), dim3(256), 0, s, (const float*)a->workspace,
                         (int)(2 * ((a->M + 255) / 256)), (int)a->N, a->colsum);
    WFT_CHECK_LAUNCH();
    if (a->colsum && !cs_fused) return wft_colsum_bf16((const wft_bf16*)a->C, a->M, a->N, a->ldc, a->colsum, 0, stream);
    return WFT_OK;
  }
  // p_valid: B is a rank-r operand in the first rows of a 128-row zero-padded buffer (u = x (sA*mask)^T, du = dy (sB)): the
  // load-stream kernel, which writes only the data columns of C (WFT_GEMM_DIAG=9: the 128-tile kernel, A/B runs)
  const int npb = (!a->c_is_f32 && a->N == 128 && a->batch == 1 && a->p_valid > 0 && a->p_valid <= 64 && a->epilogue == WFT_EPI_NONE &&
                   !a->bias && !a->residual && !a->aux && !a->colsum && a->valid_rows_period == 0 && g_diag != 9)
                      ? (a->p_valid + 15) / 16 : 0;
  if (npb) {
    const dim3 g1((unsigned)((a->M + 127) / 128));
#define WFT_NT_RANK_LAUNCH(PBV)                                                                   \
  {                                                                                               \
    constexpr int nst = (PBV) <= 2 ? 4 : 3;                                                       \
    constexpr int bytes = nst * (16384 + 2048 * (PBV));                                           \
    static DynLdsOnce once;                                                                       \
    auto kfn = gemm_nt_rank_kernel<PBV>;                                                          \
    if (!once.set(kfn, bytes)) return WFT_ERR_LAUNCH;                                                                         \
    hipLaunchKernelGGL(kfn, g1, dim3(256), bytes, s, p, p, (int)g1.x);                            \
  }
    if (npb == 1) WFT_NT_RANK_LAUNCH(1)
    else if (npb == 2) WFT_NT_RANK_LAUNCH(2)
    else if (npb == 3) WFT_NT_RANK_LAUNCH(3)
    else WFT_NT_RANK_LAUNCH(4)
#undef WFT_NT_RANK_LAUNCH
    WFT_CHECK_LAUNCH();
    return WFT_OK;
  }
  const long tiles = ((a->M + 127) / 128) * (a->N / 128);
  dim3 grid((unsigned)tiles, 1, (unsigned)a->batch), block(256);
  // a grid of at most one workgroup per CU: the four-buffer ring form (one exposed load latency per CALL instead of one per
  // k-step; WFT_GEMM_DIAG=11 keeps the two-buffer form for A/B runs)
  const bool ring = tiles * a->batch <= wft_num_cus() && g_diag != 11;
#define LAUNCH_NT_RING(E, F)                                                       \
  do {                                                                             \
    auto kfn = gemm_nt_kernel<E, F, 4>;                                            \
    static DynLdsOnce once;                                                        \
    if (!once.set(kfn, 131072)) return WFT_ERR_LAUNCH;                             \
    hipLaunchKernelGGL(kfn, grid, block, 131072, s, p);                            \
  } while (0)
  if (ring) {
    int per = 0;
    const int ns = nt_splitk_plan(a, &per);
    if (ns > 1 && a->workspace && a->workspace_bytes >= (int64_t)ns * a->M * a->N * 4 && (((uintptr_t)a->workspace) & 15) == 0) {
      GemmP ps = p;  // fp32 partial tiles [split][M][N]; alpha is applied to every partial (linear)
      ps.C = a->workspace; ps.ldc = a->N; ps.accumulate = 0; ps.nsplit = ns; ps.band = per;
      grid.z = (unsigned)ns;
      {
        auto kfn = gemm_nt_kernel<WFT_EPI_NONE, true, 4>;
        static DynLdsOnce once;
        if (!once.set(kfn, 131072)) return WFT_ERR_LAUNCH;
        hipLaunchKernelGGL(kfn, grid, block, 131072, s, ps);
      }
      hipLaunchKernelGGL(nt_splitk_reduce_kernel, dim3((unsigned)((a->M * (a->N / 4) + 255) / 256)), dim3(256), 0, s,
                         (const float*)a->workspace, ns, (int)a->M, (int)a->N, (unsigned short*)a->C, (long)a->ldc);
      WFT_CHECK_LAUNCH();
      return WFT_OK;
    }
    switch (a->epilogue) {
      case WFT_EPI_NONE: if (a->c_is_f32) LAUNCH_NT_RING(WFT_EPI_NONE, true); else LAUNCH_NT_RING(WFT_EPI_NONE, false); break;
      case WFT_EPI_GELU: if (a->c_is_f32) LAUNCH_NT_RING(WFT_EPI_GELU, true); else LAUNCH_NT_RING(WFT_EPI_GELU, false); break;
      case WFT_EPI_DGELU: if (a->c_is_f32) LAUNCH_NT_RING(WFT_EPI_DGELU, true); else LAUNCH_NT_RING(WFT_EPI_DGELU, false); break;
      case WFT_EPI_GELU_GRAD: LAUNCH_NT_RING(WFT_EPI_GELU_GRAD, false); break;
      case WFT_EPI_MUL_AUX: LAUNCH_NT_RING(WFT_EPI_MUL_AUX, false); break;
      default: wft_set_error("wft_gemm_nt_bf16: unknown epilogue %d", a->epilogue); return WFT_ERR_ARG;
    }
    WFT_CHECK_LAUNCH();
    if (a->colsum) return wft_colsum_bf16((const wft_bf16*)a->C, a->M, a->N, a->ldc, a->colsum, 0, stream);
    return WFT_OK;
  }
#undef LAUNCH_NT_RING
#define LAUNCH_NT(E)                                                               \
  do {                                                                             \
    if (a->c_is_f32) hipLaunchKernelGGL((gemm_nt_kernel<E, true>), grid, block, 0, s, p);  \
    else hipLaunchKernelGGL((gemm_nt_kernel<E, false>), grid, block, 0, s, p);     \
  } while (0)
  switch (a->epilogue) {
    case WFT_EPI_NONE: LAUNCH_NT(WFT_EPI_NONE); break;
    case WFT_EPI_GELU: LAUNCH_NT(WFT_EPI_GELU); break;
    case WFT_EPI_DGELU: LAUNCH_NT(WFT_EPI_DGELU); break;
    case WFT_EPI_GELU_GRAD: hipLaunchKernelGGL((gemm_nt_kernel<WFT_EPI_GELU_GRAD, false>), grid, block, 0, s, p); break;
    case WFT_EPI_MUL_AUX: hipLaunchKernelGGL((gemm_nt_kernel<WFT_EPI_MUL_AUX, false>), grid, block, 0, s, p); break;
    default: wft_set_error("wft_gemm_nt_bf16: unknown epilogue %d", a->epilogue); return WFT_ERR_ARG;
  }
#undef LAUNCH_NT
  WFT_CHECK_LAUNCH();
  if (a->colsum) return wft_colsum_bf16((const wft_bf16*)a->C, a->M, a->N, a->ldc, a->colsum, 0, stream);
  return WFT_OK;
}

static bool tn_uses_256(const wft_gemm_args* a) {
  const long nsteps = ((a->K + 63) / 64) * a->batch;
  // (round 5: 25 output tiles — a decoder block's 1280 x 1280 gradients — take the 256 x 256 kernel from 8 192 reduction rows on:
  // 59 -> 57 us at R = 8 704, 66 -> 56 us at R = 11 136, but 35 -> 53 us at R = 4 096; tools/dev/tn_small.py)
  const long t256 = (a->M / 256) * (a->N / 256);
  return !g_force_128 && a->c_is_f32 && a->M % 256 == 0 && a->N % 256 == 0 && nsteps >= g_tn256_min_steps &&
         (nsteps >= 256 || t256 >= g_tn256_min_out_tiles || (t256 >= g_tn256_min_out_tiles / 2 && nsteps >= 128));
}
static int tn256_nsplit(const wft_gemm_args* a) {
  const long t256 = (a->M / 256) * (a->N / 256);
  const long nslabs = ((a->K + 31) / 32) * a->batch;
  const int ncu = wft_num_cus();
  int nsplit = 1;
  double best = 0.0;
  for (int sp = 1; sp <= 16; ++sp) {
    if (sp > 1 && nslabs / sp < 48) break;
    const double waves = (double)(t256 * sp) / (double)ncu;
    const double eff = waves / (double)((long)(waves + 0.999999));
    if (eff > best + 0.02) { best = eff; nsplit = sp; }
  }
  // the kernel gives split y the slab range [y * per, (y + 1) * per), per = ceil(nslabs / nsplit): drop the splits that range
  // leaves EMPTY (they would return before storing their workspace tile, and the reduce kernel would add garbage)
  const long per = (nslabs + nsplit - 1) / nsplit;
  return (int)((nslabs + per - 1) / per);
}
// 128x128 path: split-K factor that fills the 512 resident-block slots (256 CUs x 2) in whole waves
// the ring form of gemm_tn_kernel (one workgroup per CU): the general fp32 product on a grid that fits the chip once
static bool tn128_ring(const wft_gemm_args* a) {
  // (P = 128 is the rank-r operand's buffer width: that product stays bit-identical to its p_valid form, gemm_tn_rank_kernel)
  // measured (tools/dev/small_gemm_time.py, two-buffer -> ring): R = 1 024: 16 tiles 19.2 -> 12.8 us, 48-64 tiles 19.1 -> 16.9-19.0;
  // R = 12 000: 16 tiles 28.4 -> 23.8, 32 tiles 34.3 -> 32.1, but 48 / 64 tiles 41.6 -> 43.2 / 50.4 -> 52.0 (two workgroups per CU win)
  const long tiles = (a->M / 128) * (a->N / 128), nsteps = ((a->K + 63) / 64) * a->batch;
  return a->c_is_f32 && a->M != 128 && a->tn_col_scale == nullptr && a->tn_block_n == 0 && tiles <= wft_num_cus() &&
         (tiles <= 32 || nsteps <= 32) && g_diag != 13;
}
static int tn128_nsplit(const wft_gemm_args* a) {
  if (!a->c_is_f32) return 1;
  const long tiles = (a->M / 128) * (a->N / 128);
  const long nsteps = ((a->K + 63) / 64) * a->batch;
  if (tn128_ring(a)) {
    // as many splits as fill the chip once, four reduction steps each at least (the ring's depth)
    long sp = wft_num_cus() / tiles;
    if (sp > nsteps / 4) sp = nsteps / 4;
    if (sp < 1) sp = 1;
    const long per = (nsteps + sp - 1) / sp;
    return (int)((nsteps + per - 1) / per);
  }
  int nsplit = 1;
  double best = 0.0;
  // up to 64 splits: rank-r LoRA gradients are ONE 128-wide tile row (10-40 tiles) over a 48 000+ row reduction
  for (int sp = 1; sp <= 64; ++sp) {
    if (sp > 1 && nsteps / sp < (sp <= 8 ? 16 : 12)) break;
    const double waves = (double)(tiles * sp) / 512.0;
    const double eff = waves / (double)((long)(waves + 0.999999));
    if (eff > best + 0.03) { best = eff; nsplit = sp; }
  }
  const long per = (nsteps + nsplit - 1) / nsplit;  // no empty split (see tn256_nsplit)
  return (int)((nsteps + per - 1) / per);
}
// p_valid: A is a rank-r operand in a 128-wide zero-padded buffer -> number of 16-column blocks that hold data (0: general path)
static int tn128_pb(const wft_gemm_args* a) {
  return (a->c_is_f32 && a->M == 128 && a->p_valid > 0 && a->p_valid <= 64) ? (a->p_valid + 15) / 16 : 0;
}
// rows of a split's partial tile kept in the workspace
static int64_t tn_ws_rows(const wft_gemm_args* a) {
  if (tn_uses_256(a)) return a->M;
  const int pb = tn128_pb(a);
  return pb ? 16 * pb : a->M;
}
// the adapter-gradient forms (column scale / block-transposed output) are applied by the reduce kernel: always through the workspace
static bool tn_needs_reduce(const wft_gemm_args* a) { return a->tn_col_scale != nullptr || a->tn_block_n > 0; }
extern "C" int wft_gemm_tn_segments_ok(const wft_gemm_args* a) {
  if (!a || a->tn_seg_count < 1 || a->tn_seg_count > 4 || !tn_uses_256(a) || tn_needs_reduce(a) || a->p_valid != 0 || a->batch != 1) return 0;
  int prev = 0;
  for (int i = 0; i < a->tn_seg_count; ++i) {
    if (a->tn_seg_end[i] <= prev || !a->tn_seg_ptr[i] || (((uintptr_t)a->tn_seg_ptr[i]) & 15) != 0) return 0;
    prev = a->tn_seg_end[i];
  }
  return prev == a->M ? 1 : 0;
}
static TnSegs tn_segs_of(const wft_gemm_args* a) {
  TnSegs sg;
  for (int i = 0; i < 4; ++i) {
    const int j = i < a->tn_seg_count ? i : a->tn_seg_count - 1;
    sg.ptr[i] = a->tn_seg_ptr[j];
    sg.end[i] = i < a->tn_seg_count ? a->tn_seg_end[i] : 0x7fffffff;
  }
  return sg;
}
extern "C" int64_t wft_gemm_tn_workspace_bytes(const wft_gemm_args* a) {
  if (!a) return 0;
  int nsplit = tn_uses_256(a) ? tn256_nsplit(a) : tn128_nsplit(a);
  if (a->tn_seg_count > 0) {  // segmented output: always through the workspace, even unsplit
    if (tn_uses_256(a) && wft_tn4w_eligible(a)) {
      int ns4, per4;
      wft_tn4w_plan(a, &ns4, &per4);
      if (ns4 > nsplit) nsplit = ns4;
    }
    return (int64_t)nsplit * a->M * a->N * 4;
  }
  if (tn_uses_256(a) && wft_tn4w_eligible(a)) {  // (the larger of the two plans: the variant switch may change between this call and the launch)
    int ns4, per4;
    wft_tn4w_plan(a, &ns4, &per4);
    if (ns4 > nsplit) nsplit = ns4;
  }
  return (nsplit > 1 || tn_needs_reduce(a)) ? (int64_t)nsplit * tn_ws_rows(a) * a->N * 4 : 0;
}

extern "C" int wft_gemm_tn_bf16(const wft_gemm_args* a, void* stream) {
  WFT_CHECK_ARG(a && a->A && a->B && a->C, "null pointer");
  WFT_CHECK_ARG(a->M >= 128 && a->N >= 128 && a->K >= 1 && a->batch >= 1, "bad shape");
  WFT_CHECK_ARG(a->M % 128 == 0 && a->N % 128 == 0, "P and Q must be multiples of 128");
  WFT_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 4 == 0, "ld alignment");
  WFT_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0,
                "base pointers must be 16-byte aligned");
  WFT_CHECK_ARG(!(a->accumulate && !a->c_is_f32), "accumulate needs an f32 C");
  WFT_CHECK_ARG(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "dims exceed int32");
  if (tn_needs_reduce(a)) {
    WFT_CHECK_ARG(tn128_pb(a) > 0 && !tn_uses_256(a), "tn_col_scale / tn_block_n need a rank-r operand (P = 128, 0 < p_valid <= 64)");
    WFT_CHECK_ARG(a->workspace && a->workspace_bytes >= wft_gemm_tn_workspace_bytes(a) && (((uintptr_t)a->workspace) & 15) == 0,
                  "tn_col_scale / tn_block_n need the workspace of wft_gemm_tn_workspace_bytes");
    WFT_CHECK_ARG(a->tn_scale_rows >= 0 && (((uintptr_t)a->tn_col_scale) & 15) == 0, "tn_col_scale: 16-byte aligned f32 [S][Q]");
    if (a->tn_block_n > 0)
      WFT_CHECK_ARG(a->tn_block_n % 4 == 0 && a->N % a->tn_block_n == 0 && a->tn_block_r >= 1 &&
                    (a->N / a->tn_block_n) * (int64_t)a->tn_block_r <= 16 * tn128_pb(a), "tn_block_n / tn_block_r do not tile the product");
  }
  const bool seg = a->tn_seg_count > 0;
  if (seg)
    WFT_CHECK_ARG(wft_gemm_tn_segments_ok(a) && a->workspace && a->workspace_bytes >= wft_gemm_tn_workspace_bytes(a) &&
                  (((uintptr_t)a->workspace) & 15) == 0, "tn_seg_*: not a segmentable call (wft_gemm_tn_segments_ok) or no workspace");
  GemmP p;
  fill_params(a, p);
  hipStream_t s = (hipStream_t)stream;
  const long nsteps = ((a->K + 63) / 64) * a->batch;
  if (tn_uses_256(a) && g_tn_variant != 1 && a->variant == 0 && wft_tn4w_eligible(a)) {
    int nsplit, per;
    wft_tn4w_plan(a, &nsplit, &per);
    const bool use_ws = (nsplit > 1 || seg) && a->workspace && a->workspace_bytes >= (int64_t)nsplit * a->M * a->N * 4 &&
                        (((uintptr_t)a->workspace) & 15) == 0;
    if (nsplit == 1 || use_ws) {  // (split without a workspace: the ping-pong kernel's atomic path below)
      if (use_ws) p.ws = (float*)a->workspace;
      const int rc4 = wft_tn4w_launch(a, p, nsplit, per, stream);
      if (rc4 != WFT_OK) return rc4;
      if (use_ws && seg) {
        const long total = a->M * (a->N / 4);
        long g = (total + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(tn_splitk_reduce_seg_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float*)a->workspace, tn_segs_of(a),
                           (int)a->M, (int)a->N, nsplit, a->accumulate);
      } else if (use_ws) {
        const long total = a->M * (a->N / 4);
        long g = (total + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(tn_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float*)a->workspace, (float*)a->C,
                           (long)a->ldc, (int)a->M, (int)a->N, nsplit, a->accumulate, (int)a->M, (const float*)nullptr, 0, 0, 0, (int)a->M);
      }
      WFT_CHECK_LAUNCH();
      return WFT_OK;
    }
  }
  if (tn_uses_256(a)) {
    // 256x256 tiles, one workgroup per CU: pick the split-K factor that fills 256 slots in whole waves
    const long t256 = (a->M / 256) * (a->N / 256);
    const int nsplit = tn256_nsplit(a);
    const bool use_ws = (nsplit > 1 || seg) && a->workspace && a->workspace_bytes >= (int64_t)nsplit * a->M * a->N * 4 &&
                        (((uintptr_t)a->workspace) & 15) == 0;
    if (use_ws) p.ws = (float*)a->workspace;
    if (nsplit > 1 && !use_ws && !a->accumulate)
      (void)hipMemset2DAsync(a->C, (size_t)a->ldc * 4, 0, (size_t)a->N * 4, (size_t)a->M, s);
    static DynLdsOnce once;
    auto kfn = gemm_tn256_kernel<true>;
    if (!once.set(kfn, WFT_TN_RING * 32768)) return WFT_ERR_LAUNCH;
    p.nsplit = g_diag == 20 ? -nsplit : nsplit;  // (WFT_GEMM_DIAG=20: round 2's tile-major placement, A/B runs)
    hipLaunchKernelGGL(kfn, dim3((unsigned)(t256 * nsplit)), dim3(512), WFT_TN_RING * 32768, s, p);
    if (use_ws && seg) {
      const long total = a->M * (a->N / 4);
      long g = (total + 255) / 256;
      if (g > 2048) g = 2048;
      hipLaunchKernelGGL(tn_splitk_reduce_seg_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float*)a->workspace, tn_segs_of(a),
                         (int)a->M, (int)a->N, nsplit, a->accumulate);
    } else if (use_ws) {
      const long total = a->M * (a->N / 4);
      long g = (total + 255) / 256;
      if (g > 2048) g = 2048;
      hipLaunchKernelGGL(tn_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float*)a->workspace, (float*)a->C,
                         (long)a->ldc, (int)a->M, (int)a->N, nsplit, a->accumulate, (int)a->M, (const float*)nullptr, 0, 0, 0, (int)a->M);
    }
    WFT_CHECK_LAUNCH();
    return WFT_OK;
  }
  const long tiles = (a->M / 128) * (a->N / 128);
  const int nsplit = tn128_nsplit(a);
  const bool use_ws = (nsplit > 1 || tn_needs_reduce(a)) && a->workspace &&
                      a->workspace_bytes >= (int64_t)nsplit * tn_ws_rows(a) * a->N * 4 && (((uintptr_t)a->workspace) & 15) == 0;
  if (use_ws) p.ws = (float*)a->workspace;
  if (nsplit > 1 && !use_ws && !a->accumulate)
    (void)hipMemset2DAsync(a->C, (size_t)a->ldc * 4, 0, (size_t)a->N * 4, (size_t)a->M, s);
  dim3 grid((unsigned)tiles, (unsigned)nsplit), block(256);
  // p_valid: A is a rank-r operand in a 128-wide zero-padded buffer — its own load-stream kernel (gemm_tn_rank_kernel); the
  // 128-tile kernel below only when a split-K run was given no workspace (WFT_GEMM_DIAG=9 forces it: A/B runs)
  const int pb = tn128_pb(a);
  if (pb > 0 && (nsplit == 1 || use_ws) && g_diag != 9) {
    p.nsplit = nsplit;
    const dim3 g1((unsigned)(tiles * nsplit));
#define WFT_RANK_LAUNCH(PBV)                                                                      \
  {                                                                                               \
    constexpr int nst = (PBV) <= 2 ? 4 : 3;                                                       \
    constexpr int bytes = nst * (16384 + 2048 * (PBV));                                           \
    static DynLdsOnce once;                                                                       \
    auto kfn = gemm_tn_rank_kernel<PBV>;                                                          \
    if (!once.set(kfn, bytes)) return WFT_ERR_LAUNCH;                                                                         \
    hipLaunchKernelGGL(kfn, g1, block, bytes, s, p, p, (int)g1.x);                                \
  }
    if (pb == 1) WFT_RANK_LAUNCH(1)
    else if (pb == 2) WFT_RANK_LAUNCH(2)
    else if (pb == 3) WFT_RANK_LAUNCH(3)
    else WFT_RANK_LAUNCH(4)
#undef WFT_RANK_LAUNCH
  } else if (!a->c_is_f32) hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, block, 0, s, p);
  else if (pb == 1) hipLaunchKernelGGL((gemm_tn_kernel<true, 1>), grid, block, 0, s, p);
  else if (pb == 2) hipLaunchKernelGGL((gemm_tn_kernel<true, 2>), grid, block, 0, s, p);
  else if (pb == 3) hipLaunchKernelGGL((gemm_tn_kernel<true, 3>), grid, block, 0, s, p);
  else if (pb == 4) hipLaunchKernelGGL((gemm_tn_kernel<true, 4>), grid, block, 0, s, p);
  else if (tn128_ring(a)) {
    static DynLdsOnce once;
    auto kfn = gemm_tn_kernel<true, 0, 4>;
    if (!once.set(kfn, 131072)) return WFT_ERR_LAUNCH;
    hipLaunchKernelGGL(kfn, grid, block, 131072, s, p);
  } else hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, block, 0, s, p);
  if (use_ws) {
    const long total = a->M * (a->N / 4);
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(tn_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float*)a->workspace, (float*)a->C,
                       (long)a->ldc, (int)tn_ws_rows(a), (int)a->N, nsplit, a->accumulate, (int)a->M, a->tn_col_scale,
                       a->tn_scale_rows, a->tn_block_n, a->tn_block_r, pb > 0 ? a->p_valid : (int)a->M);
  }
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}


// ---------------------------------------------------------------------------------- paired rank-r launches (round 3)
// The four rank-r products of one adapted Linear group's backward are two independent pairs: {u = x (sA*m)^T, du = dy (sB)} and
// {dA = du^T x, dB^T = u^T dy}.  Each pair goes out as ONE launch of the load-stream kernel (and the two split-K reduces of the
// second pair as one): half the launches, and a grid that fills the chip (an NT rank product alone is 1.46 rounds of 256 CUs at
// 32 clips).  Same arithmetic per product as the single entry points: bit-identical results.  Anything the load-stream kernels do
// not take falls back to two calls of the single entry point.
static int nt_rank_pb(const wft_gemm_args* a) {
  return (!a->c_is_f32 && a->N == 128 && a->batch == 1 && a->p_valid > 0 && a->p_valid <= 64 && a->epilogue == WFT_EPI_NONE &&
          !a->bias && !a->residual && !a->aux && !a->colsum && a->valid_rows_period == 0 && g_diag != 9 && a->M >= 1 && a->K >= 64 &&
          a->K % 64 == 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 4 == 0 && a->A && a->B && a->C &&
          (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) == 0 && a->M < (1ll << 31) && a->K < (1ll << 31))
             ? (a->p_valid + 15) / 16 : 0;
}
extern "C" int wft_gemm_nt_rank_pair_bf16(const wft_gemm_args* a0, const wft_gemm_args* a1, void* stream) {
  WFT_CHECK_ARG(a0 && a1, "null pointer");
  const int pb0 = nt_rank_pb(a0), pb1 = nt_rank_pb(a1);
  if (pb0 == 0 || pb0 != pb1 || g_diag == 21) {  // (WFT_GEMM_DIAG=21: always two launches, A/B runs)
    const int rc = wft_gemm_nt_bf16(a0, stream);
    return rc != WFT_OK ? rc : wft_gemm_nt_bf16(a1, stream);
  }
  GemmP p0, p1;
  fill_params(a0, p0);
  fill_params(a1, p1);
  const int n0 = (int)((a0->M + 127) / 128), n1 = (int)((a1->M + 127) / 128);
  hipStream_t s = (hipStream_t)stream;
#define WFT_NT_RANK_PAIR(PBV)                                                                     \
  {                                                                                               \
    constexpr int nst = (PBV) <= 2 ? 4 : 3;                                                       \
    constexpr int bytes = nst * (16384 + 2048 * (PBV));                                           \
    static DynLdsOnce once;                                                                       \
    auto kfn = gemm_nt_rank_kernel<PBV>;                                                          \
    if (!once.set(kfn, bytes)) return WFT_ERR_LAUNCH;                                                                         \
    hipLaunchKernelGGL(kfn, dim3((unsigned)(n0 + n1)), dim3(256), bytes, s, p0, p1, n0);          \
  }
  if (pb0 == 1) WFT_NT_RANK_PAIR(1)
  else if (pb0 == 2) WFT_NT_RANK_PAIR(2)
  else if (pb0 == 3) WFT_NT_RANK_PAIR(3)
  else WFT_NT_RANK_PAIR(4)
#undef WFT_NT_RANK_PAIR
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// both through the workspace + reduce (what the adapter-gradient forms tn_col_scale / tn_block_n always do)
static bool tn_rank_pair_ok(const wft_gemm_args* a) {
  return a->A && a->B && a->C && a->c_is_f32 && tn128_pb(a) > 0 && !tn_uses_256(a) && a->N >= 128 && a->N % 128 == 0 && a->K >= 1 &&
         a->batch >= 1 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 4 == 0 &&
         (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->workspace | (uintptr_t)a->tn_col_scale) & 15) == 0 &&
         a->workspace && a->workspace_bytes >= (int64_t)tn128_nsplit(a) * tn_ws_rows(a) * a->N * 4 && a->tn_scale_rows >= 0 &&
         (a->tn_block_n == 0 || (a->tn_block_n % 4 == 0 && a->N % a->tn_block_n == 0 && a->tn_block_r >= 1 &&
                                 (a->N / a->tn_block_n) * (int64_t)a->tn_block_r <= 16 * tn128_pb(a))) &&
         a->N < (1ll << 31) && a->K < (1ll << 31) && g_diag != 9 && g_diag != 21;
}
extern "C" int wft_gemm_tn_rank_pair_bf16(const wft_gemm_args* a0, const wft_gemm_args* a1, void* stream) {
  WFT_CHECK_ARG(a0 && a1, "null pointer");
  if (!tn_rank_pair_ok(a0) || !tn_rank_pair_ok(a1) || tn128_pb(a0) != tn128_pb(a1)) {
    const int rc = wft_gemm_tn_bf16(a0, stream);
    return rc != WFT_OK ? rc : wft_gemm_tn_bf16(a1, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const wft_gemm_args* as[2] = {a0, a1};
  GemmP p[2];
  TnReduceP r[2];
  int nb[2], gr[2];
  for (int i = 0; i < 2; ++i) {
    const wft_gemm_args* a = as[i];
    fill_params(a, p[i]);
    const int nsplit = tn128_nsplit(a);
    p[i].ws = (float*)a->workspace;
    p[i].nsplit = nsplit;
    nb[i] = (int)((a->N / 128) * nsplit);
    const long total = a->M * (a->N / 4);
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    gr[i] = (int)g;
    r[i] = TnReduceP{(const float*)a->workspace, (float*)a->C, (long)a->ldc, (int)tn_ws_rows(a), (int)a->N, nsplit, a->accumulate,
                     (int)a->M, a->tn_col_scale, a->tn_scale_rows, a->tn_block_n, a->tn_block_r, a->p_valid};
  }
  const int pb = tn128_pb(a0);
#define WFT_TN_RANK_PAIR(PBV)                                                                     \
  {                                                                                               \
    constexpr int nst = (PBV) <= 2 ? 4 : 3;                                                       \
    constexpr int bytes = nst * (16384 + 2048 * (PBV));                                           \
    static DynLdsOnce once;                                                                       \
    auto kfn = gemm_tn_rank_kernel<PBV>;                                                          \
    if (!once.set(kfn, bytes)) return WFT_ERR_LAUNCH;                                                                         \
    hipLaunchKernelGGL(kfn, dim3((unsigned)(nb[0] + nb[1])), dim3(256), bytes, s, p[0], p[1], nb[0]); \
  }
  if (pb == 1) WFT_TN_RANK_PAIR(1)
  else if (pb == 2) WFT_TN_RANK_PAIR(2)
  else if (pb == 3) WFT_TN_RANK_PAIR(3)
  else WFT_TN_RANK_PAIR(4)
#undef WFT_TN_RANK_PAIR
  hipLaunchKernelGGL(tn_splitk_reduce_pair_kernel, dim3((unsigned)(gr[0] + gr[1])), dim3(256), 0, s, r[0], r[1], gr[0]);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
