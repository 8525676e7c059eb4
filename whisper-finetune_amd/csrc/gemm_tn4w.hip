// gemm_tn4w.hip — C[P,Q] (+)= A[R,P]^T · B[R,Q], bf16 in / fp32 MFMA accumulate / fp32 out: the weight-gradient GEMM
// (dW = dY^T X of whisper.model.Linear, reached from loss.backward() in
// /root/reference/src/whisper_finetune/model/model_utils.py:83-84) as a one-wave-per-SIMD kernel, the sibling of gemm_nt4w.hip.
//
//   tile 256 (P) x 256 (Q), reduction step 64 rows of R, waves 2 (P) x 2 (Q) with 128 x 128 accumulators each in a[0:255];
//   v_mfma_f32_16x16x32_bf16 with the Q fragment as source A: a lane's 4 result registers are 4 consecutive q of one p row;
//   both operands are read ACROSS their rows (a fragment is 16 columns x 32 reduction rows): ds_read_b64_tr_b16, two per
//   fragment.  The LDS image is made for them: one LDS-DMA piece (1 KiB) = 512 B of reduction row r (lanes 0-31) and of row
//   r + 32 (lanes 32-63); pieces are 1 056 B apart and piece pi holds r = 8 g + 4 t + r_in with
//   pi = r_in | (g & 1) << 2 | t << 3 | (g >> 1) << 4, so that the 8 rows a 32-lane half reads (r_in 0..3 of two lane groups)
//   are 8 pieces whose offsets differ by multiples of 1 056 B = 8 banks + a multiple of 64: every bank once, no XOR swizzle,
//   and EVERY fragment address of a lane is ONE base register + an immediate (32 f + 8 448 t + 512 s);
//   two buffers of {A, B} x 32 pieces = 132 KiB; the same counted-vmcnt pipeline as the NT kernel (k-step t+2 loaded into the
//   half-buffer just fenced, k-step t+1's fragments read under k-step t's second half);
//   split-K: a workgroup owns (tile, split) = an even number of reduction steps; its partial tile goes to the caller's workspace
//   [split][P][Q] (summed in split order by tn_splitk_reduce_kernel: bitwise reproducible) or, unsplit, straight to C;
//   rows beyond R: the buffer descriptors end at row R, out-of-range lanes of a piece land in LDS as zeros.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

#define TN4W_BLK 1056
#define TN4W_OP (32 * TN4W_BLK)
#define TN4W_BUF (2 * TN4W_OP)
#define TN4W_LDS (2 * TN4W_BUF)  // 135 168 B

#define TN4W_STR2(x) #x
#define TN4W_STR(x) TN4W_STR2(x)

// MFMA slot s (0..127) of a reduction step: sub-step h = s / 64 (rows 0-31 / 32-63), P fragment fp = (s % 64) / 8, Q fragment
// fq = s % 8.  Fragment registers: set h: Q v[128 + 64 h + 4 fq ..+3], P v[160 + 64 h + 4 fp ..+3]; accumulator a[4 (8 fp + fq) ..+3].
//   slots  0-15       : tr-read Q set 1 (rows 32-63 of this buffer), 16 reads
//   slot  17 / 18     : lgkmcnt(0) / barrier                 -> this buffer's Q part is free
//   slots 19-34       : tr-read P set 1 (16 reads); even slots 20-34 also issue the 8 LDS-DMA pieces of Q of step t+2
//   slot  36 / 37     : lgkmcnt(0) / barrier                 -> this buffer's P part is free
//   slots 38-52 even  : LDS-DMA pieces of P of step t+2; slot 56: source bases += 64 rows, bounds -= 64 rows
//   slot  66 / 67     : vmcnt(24) / barrier                  -> step t+1's Q has landed for every wave
//   slots 68-83       : tr-read Q set 0 of step t+1
//   slot  85 / 86     : vmcnt(16) / barrier                  -> step t+1's P has landed
//   slots 87-102      : tr-read P set 0 of step t+1
// LDS-DMA pieces of step t+2: Q piece j in slot 20 + SPQ j, P piece j in slot 38 + SPP j (round 6; rounds 4-5: every second slot — see
// the note at NT4W_SPX in gemm_nt4w.hip: the pieces' issue cost depends on how crowded their phase is)
#ifndef TN4W_SPP
#define TN4W_SPP 12  // P pieces in slots 38, 50, ..., 122
#endif
#ifndef TN4W_SPQ
#define TN4W_SPQ 12  // Q pieces in slots 20, 32, ..., 104: one piece per six slots over the whole step (1 530 -> 1 310 us at 144 000 x 1 280 x 5 120)
#endif
#define TN4W_ASM_MACROS ".set TN4W_SPP, " TN4W_STR(TN4W_SPP) "\n.set TN4W_SPQ, " TN4W_STR(TN4W_SPQ) "\n" R"ASM(
.macro TN4W_MFMA s, z
  .if \z
    v_mfma_f32_16x16x32_bf16 a[4*((\s)%%64):4*((\s)%%64)+3], v[128+64*((\s)/64)+4*((\s)%%8):128+64*((\s)/64)+4*((\s)%%8)+3], v[160+64*((\s)/64)+4*(((\s)%%64)/8):160+64*((\s)/64)+4*(((\s)%%64)/8)+3], 0
  .else
    v_mfma_f32_16x16x32_bf16 a[4*((\s)%%64):4*((\s)%%64)+3], v[128+64*((\s)/64)+4*((\s)%%8):128+64*((\s)/64)+4*((\s)%%8)+3], v[160+64*((\s)/64)+4*(((\s)%%64)/8):160+64*((\s)/64)+4*(((\s)%%64)/8)+3], a[4*((\s)%%64):4*((\s)%%64)+3]
  .endif
.endm
; one LDS-DMA piece: piece j (0..7) of this wave's share; \vb = first offset VGPR (112 P / A operand, 120 Q / B operand)
.macro TN4W_DMA j, vb, srd, m0base
  .if \j == 0
    s_mov_b32 m0, \m0base
  .else
    s_add_u32 m0, \m0base, 1056*\j   ; (absolute: the two operands' pieces interleave in the step body)
  .endif
  s_nop 0
  buffer_load_dwordx4 v[\vb+\j], s[\srd:\srd+3], 0 offen lds
.endm
; next reduction step: source bases += 64 rows, bounds -= 64 rows (not below zero: a step wholly past R reads zeros)
.macro TN4W_ADVANCE
  s_add_u32 s40, s40, s54
  s_addc_u32 s41, s41, 0
  s_sub_u32 s42, s42, s54
  s_cselect_b32 s42, 0, s42
  s_add_u32 s44, s44, s55
  s_addc_u32 s45, s45, 0
  s_sub_u32 s46, s46, s55
  s_cselect_b32 s46, 0, s46
.endm
; transposed read number i (0..15) of a fragment set: fragment f = i / 2, half t = i %% 2 (4 reduction rows each) of sub-step \h
.macro TN4W_RD i, vb, addr, h
  ds_read_b64_tr_b16 v[\vb+4*((\i)/2)+2*((\i)%%2):\vb+4*((\i)/2)+2*((\i)%%2)+1], \addr offset:32*((\i)/2)+8448*((\i)%%2)+512*\h
.endm
; one reduction step.  z: first step (accumulators start from 0); ld: issue the loads of step t+2; nx: read step t+1's fragments
.macro TN4W_KSTEP z, ld, nx, vmA, vmB, rdPc, rdQc, rdPn, rdQn, mP, mQ
  .set tn4w_s, 0
  .set tn4w_iss, 0
  ; pieces of ONE step issued after its last Q piece (P pieces) / after its last P piece (Q pieces); inside a slot Q goes first
  .set tn4w_paq, 0
  .set tn4w_qap, 0
  .set tn4w_j, 0
  .rept 8
    .if (38+TN4W_SPP*tn4w_j) >= (20+TN4W_SPQ*7)
      .set tn4w_paq, tn4w_paq+1
    .endif
    .if (20+TN4W_SPQ*tn4w_j) > (38+TN4W_SPP*7)
      .set tn4w_qap, tn4w_qap+1
    .endif
    .set tn4w_j, tn4w_j+1
  .endr
  .rept 128
    .if tn4w_s < 64
      TN4W_MFMA tn4w_s, \z
    .else
      TN4W_MFMA tn4w_s, 0
    .endif
    .if tn4w_s < 16
      TN4W_RD tn4w_s, 192, \rdQc, 1
    .endif
    .if tn4w_s == 17
      s_waitcnt lgkmcnt(0)
    .endif
    .if tn4w_s == 18
      s_barrier
    .endif
    .if (tn4w_s >= 19) && (tn4w_s < 35)
      TN4W_RD tn4w_s-19, 224, \rdPc, 1
    .endif
    .if (tn4w_s >= 20) && (tn4w_s < 20+8*TN4W_SPQ) && (((tn4w_s-20) %% TN4W_SPQ) == 0)
      .if \ld
        TN4W_DMA (tn4w_s-20)/TN4W_SPQ, 120, 44, \mQ
        .set tn4w_iss, tn4w_iss+1
      .endif
    .endif
    .if tn4w_s == 36
      s_waitcnt lgkmcnt(0)
    .endif
    .if tn4w_s == 37
      s_barrier
    .endif
    .if (tn4w_s >= 38) && (tn4w_s < 38+8*TN4W_SPP) && (((tn4w_s-38) %% TN4W_SPP) == 0)
      .if \ld
        TN4W_DMA (tn4w_s-38)/TN4W_SPP, 112, 40, \mP
        .set tn4w_iss, tn4w_iss+1
      .endif
    .endif
    .if ((38+7*TN4W_SPP >= 20+7*TN4W_SPQ) && (tn4w_s == 40+7*TN4W_SPP+2)) || ((38+7*TN4W_SPP < 20+7*TN4W_SPQ) && (tn4w_s == 22+7*TN4W_SPQ+2))
      .if \ld
        TN4W_ADVANCE
      .endif
    .endif
    .if \nx
      .if tn4w_s == 66
        .if \ld
          s_waitcnt vmcnt(tn4w_paq+tn4w_iss)   ; (step t+1's P pieces behind its last Q piece + the pieces of step t+2 issued so far are younger)
        .else
          s_waitcnt vmcnt(tn4w_paq)
        .endif
      .endif
      .if tn4w_s == 67
        s_barrier
      .endif
      .if (tn4w_s >= 68) && (tn4w_s < 84)
        TN4W_RD tn4w_s-68, 128, \rdQn, 0
      .endif
      .if tn4w_s == 85
        .if \ld
          s_waitcnt vmcnt(tn4w_qap+tn4w_iss)
        .else
          s_waitcnt vmcnt(tn4w_qap)
        .endif
      .endif
      .if tn4w_s == 86
        s_barrier
      .endif
      .if (tn4w_s >= 87) && (tn4w_s < 103)
        TN4W_RD tn4w_s-87, 160, \rdPn, 0
      .endif
    .endif
    .set tn4w_s, tn4w_s+1
  .endr
  s_waitcnt lgkmcnt(0)
.endm
)ASM"

#define TN4W_ASM_PURGE R"ASM(
.purgem TN4W_MFMA
.purgem TN4W_DMA
.purgem TN4W_RD
.purgem TN4W_ADVANCE
.purgem TN4W_KSTEP
)ASM"

// literal-register clobber lists: a0..a255, v110..v255, s40..s55
#define TN4W_A8(x) "a" TN4W_STR(x##0), "a" TN4W_STR(x##1), "a" TN4W_STR(x##2), "a" TN4W_STR(x##3), "a" TN4W_STR(x##4), "a" TN4W_STR(x##5), "a" TN4W_STR(x##6), "a" TN4W_STR(x##7), "a" TN4W_STR(x##8), "a" TN4W_STR(x##9)
#define TN4W_V8(x) "v" TN4W_STR(x##0), "v" TN4W_STR(x##1), "v" TN4W_STR(x##2), "v" TN4W_STR(x##3), "v" TN4W_STR(x##4), "v" TN4W_STR(x##5), "v" TN4W_STR(x##6), "v" TN4W_STR(x##7), "v" TN4W_STR(x##8), "v" TN4W_STR(x##9)
#define TN4W_S8(x) "s" TN4W_STR(x##0), "s" TN4W_STR(x##1), "s" TN4W_STR(x##2), "s" TN4W_STR(x##3), "s" TN4W_STR(x##4), "s" TN4W_STR(x##5), "s" TN4W_STR(x##6), "s" TN4W_STR(x##7), "s" TN4W_STR(x##8), "s" TN4W_STR(x##9)
#define TN4W_CLOBBER_A                                                                                                        \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", TN4W_A8(1), TN4W_A8(2), TN4W_A8(3), TN4W_A8(4), TN4W_A8(5),     \
      TN4W_A8(6), TN4W_A8(7), TN4W_A8(8), TN4W_A8(9), TN4W_A8(10), TN4W_A8(11), TN4W_A8(12), TN4W_A8(13), TN4W_A8(14),        \
      TN4W_A8(15), TN4W_A8(16), TN4W_A8(17), TN4W_A8(18), TN4W_A8(19), TN4W_A8(20), TN4W_A8(21), TN4W_A8(22), TN4W_A8(23),    \
      TN4W_A8(24), "a250", "a251", "a252", "a253", "a254", "a255"
#define TN4W_CLOBBER_V                                                                                                        \
  TN4W_V8(11), TN4W_V8(12), TN4W_V8(13), TN4W_V8(14), TN4W_V8(15), TN4W_V8(16), TN4W_V8(17), TN4W_V8(18), TN4W_V8(19),        \
      TN4W_V8(20), TN4W_V8(21), TN4W_V8(22), TN4W_V8(23), TN4W_V8(24), "v250", "v251", "v252", "v253", "v254", "v255"
#define TN4W_CLOBBER_S TN4W_S8(4), "s50", "s51", "s52", "s53", "s54", "s55"

// The reduction loop of one (tile, split): nk reduction steps (even, >= 4) starting at the rows baseP / baseQ point at.
__device__ __forceinline__ void tn4w_kloop(unsigned rdP0, unsigned rdQ0, unsigned voP, unsigned voQ, unsigned long long baseP,
                                           unsigned long long baseQ, unsigned nrP, unsigned nrQ, unsigned lda2, unsigned ldb2,
                                           unsigned mdst, unsigned nblk) {
  asm volatile(TN4W_ASM_MACROS R"ASM(
    ; ---- per-lane source offsets of this wave's 8 pieces of each operand: piece j adds {0,1,2,3,8,9,10,11}[j] reduction rows
    v_mov_b32 v112, %[voP]
    v_add_u32 v113, %[lda2], v112
    v_add_u32 v114, %[lda2], v113
    v_add_u32 v115, %[lda2], v114
    s_lshl_b32 s49, %[lda2], 3
    v_add_u32 v116, s49, v112
    v_add_u32 v117, s49, v113
    v_add_u32 v118, s49, v114
    v_add_u32 v119, s49, v115
    v_mov_b32 v120, %[voQ]
    v_add_u32 v121, %[ldb2], v120
    v_add_u32 v122, %[ldb2], v121
    v_add_u32 v123, %[ldb2], v122
    s_lshl_b32 s49, %[ldb2], 3
    v_add_u32 v124, s49, v120
    v_add_u32 v125, s49, v121
    v_add_u32 v126, s49, v122
    v_add_u32 v127, s49, v123
    ; fragment read addresses in buffer 1
    v_add_u32 v110, )ASM" TN4W_STR(TN4W_BUF) R"ASM(, %[rdP0]
    v_add_u32 v111, )ASM" TN4W_STR(TN4W_BUF) R"ASM(, %[rdQ0]
    ; ---- LDS-DMA destinations of this wave: P / Q part of buffer 0 / 1; 64 reduction rows in bytes
    s_mov_b32 s50, %[mdst]
    s_add_u32 s51, s50, )ASM" TN4W_STR(TN4W_BUF) R"ASM(
    s_add_u32 s52, s50, )ASM" TN4W_STR(TN4W_OP) R"ASM(
    s_add_u32 s53, s51, )ASM" TN4W_STR(TN4W_OP) R"ASM(
    s_lshl_b32 s54, %[lda2], 6
    s_lshl_b32 s55, %[ldb2], 6
    s_mov_b64 s[40:41], %[baseP]
    s_mov_b32 s42, %[nrP]
    s_mov_b32 s43, 0x20000
    s_mov_b64 s[44:45], %[baseQ]
    s_mov_b32 s46, %[nrQ]
    s_mov_b32 s47, 0x20000
    s_mov_b32 s48, %[nblk]
    s_nop 4
    ; ---- steps 0 and 1 into buffers 0 and 1
    .irp j,0,1,2,3,4,5,6,7
      TN4W_DMA \j, 120, 44, s52
    .endr
    .irp j,0,1,2,3,4,5,6,7
      TN4W_DMA \j, 112, 40, s50
    .endr
    TN4W_ADVANCE
    s_nop 4
    .irp j,0,1,2,3,4,5,6,7
      TN4W_DMA \j, 120, 44, s53
    .endr
    .irp j,0,1,2,3,4,5,6,7
      TN4W_DMA \j, 112, 40, s51
    .endr
    TN4W_ADVANCE
    s_waitcnt vmcnt(16)   ; step 0 complete, step 1's 16 pieces are the youngest operations
    s_barrier
    .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
      TN4W_RD \i, 128, %[rdQ0], 0
    .endr
    .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
      TN4W_RD \i, 160, %[rdP0], 0
    .endr
    s_waitcnt lgkmcnt(0)
    TN4W_KSTEP 1, 1, 1, 24, 16, %[rdP0], %[rdQ0], v110, v111, s50, s52
    TN4W_KSTEP 0, 1, 1, 24, 16, v110, v111, %[rdP0], %[rdQ0], s51, s53
    s_cmp_eq_u32 s48, 0
    s_cbranch_scc1 4f
3:
    TN4W_KSTEP 0, 1, 1, 24, 16, %[rdP0], %[rdQ0], v110, v111, s50, s52
    TN4W_KSTEP 0, 1, 1, 24, 16, v110, v111, %[rdP0], %[rdQ0], s51, s53
    s_sub_u32 s48, s48, 1
    s_cmp_eq_u32 s48, 0
    s_cbranch_scc0 3b
4:
    ; ---- last two steps: nothing to load; step nk-1's 16 pieces are the youngest operations (exact waits 8 / 0)
    TN4W_KSTEP 0, 0, 1, 8, 0, %[rdP0], %[rdQ0], v110, v111, s50, s52
    TN4W_KSTEP 0, 0, 0, 0, 0, v110, v111, %[rdP0], %[rdQ0], s51, s53
    s_nop 15
  )ASM" TN4W_ASM_PURGE
               :
               : [rdP0] "v"(rdP0), [rdQ0] "v"(rdQ0), [voP] "v"(voP), [voQ] "v"(voQ), [baseP] "s"(baseP), [baseQ] "s"(baseQ),
                 [nrP] "s"(nrP), [nrQ] "s"(nrQ), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [mdst] "s"(mdst), [nblk] "s"(nblk)
               : "memory", "vcc", "scc", TN4W_CLOBBER_A, TN4W_CLOBBER_V, TN4W_CLOBBER_S);
}

template <int N>
__device__ __forceinline__ float tn4w_acc() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "i"(N));
  return x;
}
template <int I, int N, class F>
__device__ __forceinline__ void tn4w_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tn4w_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ unsigned tn4w_sgpr(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long tn4w_sgpr64(unsigned long long x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void gemm_tn4w_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wq = wave & 1;
  const int P = p.M, Q = p.N, R = p.K;
  const int tiles_q = Q >> 8, tiles_p = P >> 8;
  const int ntile = tiles_p * tiles_q;
  const int nsplit = p.nsplit;
  // (split, tile) pairs split-major through the XCD map (the ~32 workgroups an XCD runs together are tiles of one split: they
  // walk the same reduction range in step and share its slabs in that L2 — same placement as gemm_tn256_kernel)
  const int wsid = xcd_remap(blockIdx.x, ntile * nsplit);
  const int split = wsid / ntile, sid = wsid - split * ntile;
  int tp, tq;
  band_coords(sid, tiles_p, tiles_q, tp, tq);
  const int p0 = tp << 8, q0 = tq << 8;
  // reduction steps of this split: `per` (even) steps from kb; the whole range is padded to an even count (a step past R reads
  // zeros through the descriptors' bounds)
  const int per = p.band;  // (host: even, >= 4; every split is non-empty)
  const int kb = split * per;
  const int nk_all = (((R + 63) >> 6) + 1) & ~1;
  const int nk = (kb + per <= nk_all) ? per : nk_all - kb;
  const unsigned lds0 = lds_addr_of(dsmem);
  const unsigned lda2 = tn4w_sgpr((unsigned)p.lda * 2u), ldb2 = tn4w_sgpr((unsigned)p.ldb * 2u);

  // fragment read address: lane (g = lane >> 4, li) supplies row r_in = li >> 2, columns 4 (li & 3) .. +3 of a 4 x 16 block
  const int g = lane >> 4, li = lane & 15;
  const unsigned rd0 = lds0 + (unsigned)((li >> 2) + 4 * (g & 1) + 16 * (g >> 1)) * TN4W_BLK + 8u * (li & 3);
  const unsigned rdP0 = rd0 + wp * 256, rdQ0 = rd0 + TN4W_OP + wq * 256;
  // LDS-DMA share of this wave: pieces 8 wave + j; lane: half h = lane >> 5 (rows + 32), 16-byte chunk lane & 31
  const int h = lane >> 5, cp = lane & 31;
  const int rw = 32 * h + 16 * (wave >> 1) + 4 * (wave & 1);
  const unsigned voP = (unsigned)(rw * (int)p.lda + cp * 8) * 2u, voQ = (unsigned)(rw * (int)p.ldb + cp * 8) * 2u;
  const unsigned long long baseP = tn4w_sgpr64((unsigned long long)(p.A + (long)kb * 64 * p.lda + p0));
  const unsigned long long baseQ = tn4w_sgpr64((unsigned long long)(p.B + (long)kb * 64 * p.ldb + q0));
  const long rem = (long)R - (long)kb * 64;  // valid rows from this split's first one (> 0)
  const unsigned nrP = tn4w_sgpr((unsigned)(rem * p.lda * 2)), nrQ = tn4w_sgpr((unsigned)(rem * p.ldb * 2));
  tn4w_kloop(rdP0, rdQ0, voP, voQ, baseP, baseQ, nrP, nrQ, lda2, ldb2, tn4w_sgpr(lds0 + wave * 8 * TN4W_BLK),
             tn4w_sgpr((unsigned)(nk - 4) >> 1));

  // ---- epilogue: lane (g, li) holds, for P fragment fp and Q fragment fq, C[p0 + wp*128 + 16 fp + li][q0 + wq*128 + 16 fq + 4 g .. +3]
  float* const dst = p.ws ? p.ws + ((long)split * P + p0) * Q + q0 : (float*)p.C + (long)p0 * p.ldc + q0;
  const long ldd = p.ws ? (long)Q : p.ldc;
  const bool acc_c = !p.ws && p.accumulate;
  tn4w_for<0, 8>([&](auto fpc) {
    constexpr int fp = decltype(fpc)::value;
    float* const row = dst + (long)(wp * 128 + 16 * fp + li) * ldd + wq * 128 + 4 * g;
    tn4w_for<0, 8>([&](auto fqc) {
      constexpr int fq = decltype(fqc)::value, i0 = (fp * 8 + fq) * 4;
      f32x4 o = {tn4w_acc<i0>(), tn4w_acc<i0 + 1>(), tn4w_acc<i0 + 2>(), tn4w_acc<i0 + 3>()};
      o *= p.alpha;
      if (acc_c) o += *(const f32x4*)(row + 16 * fq);
      *(f32x4*)(row + 16 * fq) = o;
    });
  });
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// Eligibility beyond tn_uses_256 (gemm.hip): one batch item, fp32 C, 32-bit byte offsets over the whole reduction range.
bool wft_tn4w_eligible(const wft_gemm_args* a) {
  return a->c_is_f32 && a->batch == 1 && a->M % 256 == 0 && a->N % 256 == 0 && a->K >= 1024 && a->p_valid == 0 && !a->tn_col_scale &&
         a->tn_block_n == 0 && a->lda >= a->M && a->ldb >= a->N && (a->K + 64) * a->lda * 2 < (1ll << 32) &&
         (a->K + 64) * a->ldb * 2 < (1ll << 32);
}
// split-K plan: nsplit splits of `per` (even) reduction steps each, every split non-empty
void wft_tn4w_plan(const wft_gemm_args* a, int* nsplit_out, int* per_out) {
  const long t256 = (a->M / 256) * (a->N / 256);
  const long nk = ((((a->K + 63) / 64) + 1) / 2) * 2;  // padded to even
  const int ncu = wft_num_cus();
  int nsplit = 1;
  double best = 0.0;
  for (int sp = 1; sp <= 16; ++sp) {
    if (sp > 1 && nk / sp < 24) break;
    const double waves = (double)(t256 * sp) / (double)ncu;
    const double eff = waves / (double)((long)(waves + 0.999999));
    if (eff > best + 0.02) { best = eff; nsplit = sp; }
  }
  long per = (nk + nsplit - 1) / nsplit;
  per = (per + 1) & ~1l;
  if (per < 4) per = 4;
  long ns = (nk + per - 1) / per;  // (no empty split)
  while (ns > 1 && nk - (ns - 1) * per < 4) {  // the K loop needs >= 4 steps: fold a 2-step remainder into longer splits
    per += 2;
    ns = (nk + per - 1) / per;
  }
  *per_out = (int)per;
  *nsplit_out = (int)ns;
}
int wft_tn4w_launch(const wft_gemm_args* a, GemmP p, int nsplit, int per, void* stream) {
  const long t256 = (a->M / 256) * (a->N / 256);
  p.nsplit = nsplit;
  p.band = per;  // (reused field: reduction steps per split)
  static DynLdsOnce once;
  auto kfn = gemm_tn4w_kernel;
  if (!once.set(kfn, TN4W_LDS)) return WFT_ERR_LAUNCH;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(t256 * nsplit)), dim3(256), TN4W_LDS, (hipStream_t)stream, p);
  return WFT_OK;
}
