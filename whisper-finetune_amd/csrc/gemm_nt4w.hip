// gemm_nt4w.hip — C[M,N] = A[M,K] · B[N,K]^T, bf16 in / fp32 MFMA accumulate / bf16 out: the one-wave-per-SIMD kernel.
//
// Replaces whisper.model.Linear's matmul on the encoder-sized problems (reached from
// /root/reference/src/whisper_finetune/model/model_utils.py:283-285,320-325).  Same C ABI entry as the ping-pong kernel
// (wft_gemm_nt_bf16 dispatches here; WFT_NT_VARIANT=pp keeps the old kernel).
//
// Why a second 256x256 kernel: the 8-wave ping-pong kernel (gemm.hip) re-reads every LDS fragment for 4-8 MFMAs and keeps two
// waves per SIMD busy issuing loads beside each other's MFMAs; rocprof shows its matrix pipe 51-54 % busy at the board's power
// cap.  Here each of FOUR waves (one per SIMD) owns a 128 x 128 block of the tile in 256 accumulation registers (a[0:255]) and
// every fragment it reads from LDS feeds 8 MFMAs: 64 B of LDS traffic per clock per CU instead of 96, a third less LDS energy at
// the 1 400 W cap, and no second wave to arbitrate with.  The price is that nothing hides a stalled instruction, so the whole K
// loop is ONE hand-scheduled inline-asm block: every ds_read / LDS-DMA / wait / barrier sits in a chosen MFMA gap.
//
//   tile 256 x 256, k-step 64, waves 2 (M) x 2 (N), MFMA v_mfma_f32_16x16x32_bf16 with the weight fragment as source A so that
//   a lane's 4 result registers are 4 consecutive output columns;
//   LDS: two k-step buffers of {X [256 rows][64 k], W [256 rows][64 k]} as 1 056-byte blocks (1 KiB data = the 8 fragment rows
//   one lane group reads, + 32 B pad: the 16 lanes of a ds_read_b128 group land on 16 different bank quads) = 132 KiB;
//   one LDS-DMA instruction (buffer_load_dwordx4 ... lds) fills one block: 8 rows x 128 B, whole lines of A / B;
//   the ROW ORDER inside the image is free (the DMA source address is per lane): W rows are dealt so that the 8 fragments of a
//   wave give each lane 8 CONSECUTIVE output columns per fragment pair -> 16-byte stores straight from registers;
//   pipeline: k-step t+2 is loaded into the half-buffer whose last fragment read has just been fenced by a barrier, k-step
//   t+1's fragments are read during the second half of k-step t's MFMAs (vmcnt-counted, never drained); the last two k-steps of
//   a tile load the NEXT tile's first two k-steps, so the persistent tile sequence is one uninterrupted load stream and the
//   epilogue runs while they land.
//
// The K loop is written with assembler macros (.macro / .rept over the 128 MFMA slots of a k-step); the slot numbers in
// NT4W_SCHED_* are the schedule.  Registers named literally in the asm (a[0:255], v[112:255], s[40:53], s[56:59]) are clobbers of the
// one statement; the epilogue reads the accumulators back with v_accvgpr_read (cdna_hip_programming.md §5.7 item 4).
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

#define NT4W_BLK 1056                 // bytes per LDS block (1 KiB + 32 B pad)
#define NT4W_OP (32 * NT4W_BLK)       // one operand of one k-step: 32 blocks
#define NT4W_BUF (2 * NT4W_OP)        // one k-step buffer (X then W)
#define NT4W_BIAS (2 * NT4W_BUF)      // two 1 KiB bias slices (this tile / next tile) behind the k-step buffers
#define NT4W_LDS (NT4W_BIAS + 2048)   // 137 216 B

// Cache policy of the output stores: sc0 nt (3).  A tile's 128 KiB of C are not read again by this kernel; written with the default
// policy they displace operand panels from the XCD's 4 MiB L2 (measured at 102000 x 5120 x 1280: default 1 184-1 189 TF/s, nt
// 1 219, sc0 nt 1 230, sc1 1 218; WFT_GEMM_DIAG=30 restores the default for A/B runs).
#define NT4W_ST_AUX 3

#define NT4W_STR2(x) #x
#define NT4W_STR(x) NT4W_STR2(x)

// ---- the main loop ------------------------------------------------------------------------------------------------------
// MFMA slot s (0..127) of a k-step: sub-step h = s / 64 (k 0-31 / 32-63), X fragment fx = (s % 64) / 8, W fragment fw = s % 8.
//   fragment registers: set h: W  v[128 + 64 h + 4 fw ..+3],  X  v[160 + 64 h + 4 fx ..+3];  accumulator a[4 (8 fx + fw) ..+3]
// Per k-step (buffer c = this k-step, n = the other buffer):
//   slots  0-14 even : ds_read W set 1 (k 32-63 of c)                     8 reads
//   slot  17 / 18    : lgkmcnt(0) / barrier            -> every wave has read all of c's W part
//   slots 20-35      : even: LDS-DMA W of k-step t+2 -> c (8 pieces); odd: ds_read X set 1 (8 reads)
//   slot  37 / 38    : lgkmcnt(0) / barrier            -> c's X part is free
//   slots 40-54 even : LDS-DMA X of k-step t+2 -> c (8 pieces); slot 56: both source bases += 128 B
//   slot  66 / 67    : vmcnt(24) / barrier             -> k-step t+1's W (in n) has landed for every wave
//   slots 68-82 even : ds_read W set 0 of k-step t+1
//   slot  84 / 85    : vmcnt(16) / barrier             -> k-step t+1's X has landed
//   slots 86-100 even: ds_read X set 0 of k-step t+1
//   after slot 127   : lgkmcnt(0)
// vmcnt retires in issue order and every k-step issues 8 W pieces then 8 X pieces, so "k-step t+1's W complete" = at most 8 (its
// X) + 16 (k-step t+2) younger operations outstanding.
#ifndef NT4W_T1
#define NT4W_T1 17
#define NT4W_T2 37
#define NT4W_T3 66
#define NT4W_T4 84
#endif
// LDS-DMA pieces of k-step t+2: W piece j in slot T1+2 + SPW j, X piece j in slot T2+2 + SPX j (round 6).  Rounds 4-5 issued them in every
// second slot (19-33, 39-53): a piece costs the wave 100-185 issue cycles inside a phase that already carries 8 pieces and 16
// ds_reads and 25-60 in a quiet phase (MI355X_MICROARCH.md), and sixteen of them bunched into 36 slots were a third of the k-step's
// idle matrix-pipe time.  Spread over the k-step: +3 % (NT) / +7 % (TN) at kernel level, -3 % of the headline step.
#ifndef NT4W_SPX
#define NT4W_SPX 8   // X pieces in slots 39, 47, ..., 95 (measured: 6-11 within 1 %; 2 = rounds 4-5)
#endif
#ifndef NT4W_W0
#define NT4W_W0 (NT4W_T1+2)   // slot of W piece 0 (>= T1+2: behind the barrier that frees the W half; 36 / 42 — behind the X set-1 reads — measured 6-12 % SLOWER: the W pieces are the first thing the next k-step waits for)
#endif
#ifndef NT4W_SPW
#define NT4W_SPW 3   // W pieces in slots 19, 22, ..., 40 (2-4 equal; 6-12 measured 2-3 % slower than 2-4)
#endif
#define NT4W_ASM_MACROS ".set NT4W_T1, " NT4W_STR(NT4W_T1) "\n.set NT4W_T2, " NT4W_STR(NT4W_T2) "\n.set NT4W_T3, " NT4W_STR(NT4W_T3) "\n.set NT4W_T4, " NT4W_STR(NT4W_T4) "\n.set NT4W_SPX, " NT4W_STR(NT4W_SPX) "\n.set NT4W_SPW, " NT4W_STR(NT4W_SPW) "\n.set NT4W_W0, " NT4W_STR(NT4W_W0) "\n" R"ASM(
.macro NT4W_MFMA s, z
  .if \z
    v_mfma_f32_16x16x32_bf16 a[4*((\s)%%64):4*((\s)%%64)+3], v[128+64*((\s)/64)+4*((\s)%%8):128+64*((\s)/64)+4*((\s)%%8)+3], v[160+64*((\s)/64)+4*(((\s)%%64)/8):160+64*((\s)/64)+4*(((\s)%%64)/8)+3], 0
  .else
    v_mfma_f32_16x16x32_bf16 a[4*((\s)%%64):4*((\s)%%64)+3], v[128+64*((\s)/64)+4*((\s)%%8):128+64*((\s)/64)+4*((\s)%%8)+3], v[160+64*((\s)/64)+4*(((\s)%%64)/8):160+64*((\s)/64)+4*(((\s)%%64)/8)+3], a[4*((\s)%%64):4*((\s)%%64)+3]
  .endif
.endm
; one LDS-DMA piece: block j (0..7) of this wave's share; \vb = first offset VGPR (112 X, 120 W), \srd = first SGPR of the descriptor
.macro NT4W_DMA j, vb, srd, m0base
  .if \j == 0
    s_mov_b32 m0, \m0base
  .else
    s_add_u32 m0, \m0base, 1056*\j   ; (absolute: the two operands' pieces interleave in the k-step body)
  .endif
  s_nop 0
  buffer_load_dwordx4 v[\vb+\j], s[\srd:\srd+3], 0 offen lds
.endm
; the 256 bias values of a tile's columns -> LDS (one 1 KiB piece, issued by wave 0 only; \go = 0: no bias / not wave 0)
.macro NT4W_BIAS_DMA go, base, dst, voff
  s_cmp_eq_u32 \go, 0
  s_cbranch_scc1 9f
  s_mov_b64 s[56:57], \base
  s_mov_b32 s58, 1024
  s_mov_b32 s59, 0x20000
  s_mov_b32 m0, \dst
  s_nop 4
  buffer_load_dwordx4 \voff, s[56:59], 0 offen lds
9:
.endm
; one k-step.  z: first k-step of a tile (accumulators start from 0); ld: issue the loads of k-step t+2; nx: read k-step t+1's
; fragments (with vmA / vmB = vmcnt before its W / X reads); rdXc.. : address VGPRs of this / the other buffer; mX, mW: SGPRs with
; this wave's LDS-DMA destination in this buffer
.macro NT4W_KSTEP z, ld, nx, vmA, vmB, rdXc, rdWc, rdXn, rdWn, mX, mW
  .set nt4w_s, 0
  .set nt4w_iss, 0
  ; pieces of ONE k-step issued after its last W piece (X pieces) / after its last X piece (W pieces): what is younger than "all of
  ; k-step t+1's W" / "all of its X" besides the pieces of k-step t+2 issued so far
  .set nt4w_xaw, 0
  .set nt4w_wax, 0
  .set nt4w_j, 0
  .rept 8
    .if (NT4W_T2+2+NT4W_SPX*nt4w_j) >= (NT4W_W0+NT4W_SPW*7)   ; (inside a slot the W piece is issued first)
      .set nt4w_xaw, nt4w_xaw+1
    .endif
    .if (NT4W_W0+NT4W_SPW*nt4w_j) > (NT4W_T2+2+NT4W_SPX*7)
      .set nt4w_wax, nt4w_wax+1
    .endif
    .set nt4w_j, nt4w_j+1
  .endr
  .rept 128
    .if nt4w_s < 64
      NT4W_MFMA nt4w_s, \z
    .else
      NT4W_MFMA nt4w_s, 0
    .endif
    .if (nt4w_s < 16) && ((nt4w_s %% 2) == 0)
      ds_read_b128 v[192+4*(nt4w_s/2):192+4*(nt4w_s/2)+3], \rdWc offset:128*(nt4w_s/2)+64
    .endif
    .if nt4w_s == NT4W_T1
      s_waitcnt lgkmcnt(0)
    .endif
    .if nt4w_s == NT4W_T1+1
      s_barrier
    .endif
    .if (nt4w_s >= NT4W_W0) && (nt4w_s < NT4W_W0+8*NT4W_SPW) && (((nt4w_s-NT4W_W0) %% NT4W_SPW) == 0)
      .if \ld
        NT4W_DMA (nt4w_s-NT4W_W0)/NT4W_SPW, 120, 44, \mW
        .set nt4w_iss, nt4w_iss+1
      .endif
    .endif
    .if (nt4w_s >= NT4W_T1+3) && (nt4w_s < NT4W_T1+18) && (((nt4w_s-NT4W_T1) %% 2) == 1)
      ds_read_b128 v[224+4*((nt4w_s-NT4W_T1-3)/2):224+4*((nt4w_s-NT4W_T1-3)/2)+3], \rdXc offset:128*((nt4w_s-NT4W_T1-3)/2)+64
    .endif
    .if nt4w_s == NT4W_T2
      s_waitcnt lgkmcnt(0)
    .endif
    .if nt4w_s == NT4W_T2+1
      s_barrier
    .endif
    .if (nt4w_s >= NT4W_T2+2) && (nt4w_s < NT4W_T2+2+8*NT4W_SPX) && (((nt4w_s-NT4W_T2-2) %% NT4W_SPX) == 0)
      .if \ld
        NT4W_DMA (nt4w_s-NT4W_T2-2)/NT4W_SPX, 112, 40, \mX
        .set nt4w_iss, nt4w_iss+1
      .endif
    .endif
    .if ((NT4W_T2+2+7*NT4W_SPX >= NT4W_W0+7*NT4W_SPW) && (nt4w_s == NT4W_T2+4+7*NT4W_SPX)) || ((NT4W_T2+2+7*NT4W_SPX < NT4W_W0+7*NT4W_SPW) && (nt4w_s == NT4W_W0+2+7*NT4W_SPW))
      .if \ld
        s_add_u32 s40, s40, 128
        s_addc_u32 s41, s41, 0
        s_add_u32 s44, s44, 128
        s_addc_u32 s45, s45, 0
      .endif
    .endif
    .if \nx
      .if nt4w_s == NT4W_T3
        .if \ld
          s_waitcnt vmcnt(nt4w_xaw+nt4w_iss)   ; (k-step t+1's X pieces behind its last W piece + the pieces of k-step t+2 issued so far are younger)
        .else
          s_waitcnt vmcnt(nt4w_xaw)   ; (nothing issued in this k-step: only k-step t+1's own later pieces are younger)
        .endif
      .endif
      .if nt4w_s == NT4W_T3+1
        s_barrier
      .endif
      .if (nt4w_s >= NT4W_T3+2) && (nt4w_s < NT4W_T3+18) && (((nt4w_s-NT4W_T3) %% 2) == 0)
        ds_read_b128 v[128+4*((nt4w_s-NT4W_T3-2)/2):128+4*((nt4w_s-NT4W_T3-2)/2)+3], \rdWn offset:128*((nt4w_s-NT4W_T3-2)/2)
      .endif
      .if nt4w_s == NT4W_T4
        .if \ld
          s_waitcnt vmcnt(nt4w_wax+nt4w_iss)
        .else
          s_waitcnt vmcnt(nt4w_wax)
        .endif
      .endif
      .if nt4w_s == NT4W_T4+1
        s_barrier
      .endif
      .if (nt4w_s >= NT4W_T4+2) && (nt4w_s < NT4W_T4+18) && (((nt4w_s-NT4W_T4) %% 2) == 0)
        ds_read_b128 v[160+4*((nt4w_s-NT4W_T4-2)/2):160+4*((nt4w_s-NT4W_T4-2)/2)+3], \rdXn offset:128*((nt4w_s-NT4W_T4-2)/2)
      .endif
    .endif
    .set nt4w_s, nt4w_s+1
  .endr
  s_waitcnt lgkmcnt(0)
.endm
)ASM"

#define NT4W_ASM_PURGE R"ASM(
.purgem NT4W_MFMA
.purgem NT4W_DMA
.purgem NT4W_BIAS_DMA
.purgem NT4W_KSTEP
)ASM"

// literal-register clobber lists
// a0..a255, v112..v255, s40..s59
#define NT4W_A8(x) "a" NT4W_STR(x##0), "a" NT4W_STR(x##1), "a" NT4W_STR(x##2), "a" NT4W_STR(x##3), "a" NT4W_STR(x##4), "a" NT4W_STR(x##5), "a" NT4W_STR(x##6), "a" NT4W_STR(x##7), "a" NT4W_STR(x##8), "a" NT4W_STR(x##9)
#define NT4W_V8(x) "v" NT4W_STR(x##0), "v" NT4W_STR(x##1), "v" NT4W_STR(x##2), "v" NT4W_STR(x##3), "v" NT4W_STR(x##4), "v" NT4W_STR(x##5), "v" NT4W_STR(x##6), "v" NT4W_STR(x##7), "v" NT4W_STR(x##8), "v" NT4W_STR(x##9)
#define NT4W_S8(x) "s" NT4W_STR(x##0), "s" NT4W_STR(x##1), "s" NT4W_STR(x##2), "s" NT4W_STR(x##3), "s" NT4W_STR(x##4), "s" NT4W_STR(x##5), "s" NT4W_STR(x##6), "s" NT4W_STR(x##7), "s" NT4W_STR(x##8), "s" NT4W_STR(x##9)
#define NT4W_CLOBBER_A                                                                                                        \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", NT4W_A8(1), NT4W_A8(2), NT4W_A8(3), NT4W_A8(4), NT4W_A8(5),     \
      NT4W_A8(6), NT4W_A8(7), NT4W_A8(8), NT4W_A8(9), NT4W_A8(10), NT4W_A8(11), NT4W_A8(12), NT4W_A8(13), NT4W_A8(14),        \
      NT4W_A8(15), NT4W_A8(16), NT4W_A8(17), NT4W_A8(18), NT4W_A8(19), NT4W_A8(20), NT4W_A8(21), NT4W_A8(22), NT4W_A8(23),    \
      NT4W_A8(24), "a250", "a251", "a252", "a253", "a254", "a255"
#define NT4W_CLOBBER_V                                                                                                        \
  "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", NT4W_V8(12), NT4W_V8(13), NT4W_V8(14), NT4W_V8(15),         \
      NT4W_V8(16), NT4W_V8(17), NT4W_V8(18), NT4W_V8(19), NT4W_V8(20), NT4W_V8(21), NT4W_V8(22), NT4W_V8(23), NT4W_V8(24),    \
      "v250", "v251", "v252", "v253", "v254", "v255"
#define NT4W_CLOBBER_S NT4W_S8(4), "s50", "s51", "s52", "s53", "s56", "s57", "s58", "s59"

// Everything the K loop of one tile needs.  Wave-uniform members are SGPR operands (the caller pins them with readfirstlane).
struct Nt4wTile {
  unsigned long long baseX, baseW;  // first row of the tile's A rows / B rows, k = 0
  unsigned nrX, nrW;                // buffer bounds in bytes from the base: rows beyond M read as zero
};

template <bool DUMMY = false>
__device__ __forceinline__ void nt4w_kloop(unsigned rdX0, unsigned rdX1, unsigned rdW0, unsigned rdW1, unsigned voX, unsigned voW,
                                           const Nt4wTile& cur, const Nt4wTile& nxt, unsigned lda2, unsigned ldb2, unsigned mdst,
                                           unsigned nblk, unsigned first, unsigned more, unsigned nops, unsigned lane16, unsigned bgo,
                                           unsigned long long cbias, unsigned long long nbias, unsigned cbdst, unsigned nbdst) {
  asm volatile(NT4W_ASM_MACROS R"ASM(
    ; ---- per-lane source offsets of this wave's 8 X blocks and 8 W blocks (row step of block j: j rows of A; {0,1,2,3,8,9,10,11} rows of B)
    v_mov_b32 v112, %[voX]
    v_add_u32 v113, %[lda2], v112
    v_add_u32 v114, %[lda2], v113
    v_add_u32 v115, %[lda2], v114
    v_add_u32 v116, %[lda2], v115
    v_add_u32 v117, %[lda2], v116
    v_add_u32 v118, %[lda2], v117
    v_add_u32 v119, %[lda2], v118
    s_lshl_b32 s49, %[ldb2], 3
    v_mov_b32 v120, %[voW]
    v_add_u32 v121, %[ldb2], v120
    v_add_u32 v122, %[ldb2], v121
    v_add_u32 v123, %[ldb2], v122
    v_add_u32 v124, s49, v120
    v_add_u32 v125, s49, v121
    v_add_u32 v126, s49, v122
    v_add_u32 v127, s49, v123
    ; ---- LDS-DMA destinations of this wave: X / W part of buffer 0 / 1
    s_mov_b32 s50, %[mdst]
    s_add_u32 s51, s50, )ASM" NT4W_STR(NT4W_BUF) R"ASM(
    s_add_u32 s52, s50, )ASM" NT4W_STR(NT4W_OP) R"ASM(
    s_add_u32 s53, s51, )ASM" NT4W_STR(NT4W_OP) R"ASM(
    ; ---- buffer descriptors of the current tile
    s_mov_b64 s[40:41], %[cbX]
    s_mov_b32 s42, %[cnX]
    s_mov_b32 s43, 0x20000
    s_mov_b64 s[44:45], %[cbW]
    s_mov_b32 s46, %[cnW]
    s_mov_b32 s47, 0x20000
    s_mov_b32 s48, %[nblk]
    s_cmp_eq_u32 %[first], 0
    s_cbranch_scc1 1f
    ; first tile of this workgroup: nobody staged its bias slice and its k-steps 0 and 1
    NT4W_BIAS_DMA %[bgo], %[cbias], %[cbdst], %[lane16]
    .irp j,0,1,2,3,4,5,6,7
      NT4W_DMA \j, 120, 44, s52
    .endr
    .irp j,0,1,2,3,4,5,6,7
      NT4W_DMA \j, 112, 40, s50
    .endr
    s_add_u32 s40, s40, 128
    s_addc_u32 s41, s41, 0
    s_add_u32 s44, s44, 128
    s_addc_u32 s45, s45, 0
    .irp j,0,1,2,3,4,5,6,7
      NT4W_DMA \j, 120, 44, s53
    .endr
    .irp j,0,1,2,3,4,5,6,7
      NT4W_DMA \j, 112, 40, s51
    .endr
    s_add_u32 s40, s40, 128
    s_addc_u32 s41, s41, 0
    s_add_u32 s44, s44, 128
    s_addc_u32 s45, s45, 0
    s_waitcnt vmcnt(16)   ; k-step 0 complete, k-step 1's 16 pieces are the youngest operations
    s_branch 7f
1:
    ; k-steps 0 and 1 were staged by the previous tile's last two k-steps
    s_add_u32 s40, s40, 256
    s_addc_u32 s41, s41, 0
    s_add_u32 s44, s44, 256
    s_addc_u32 s45, s45, 0
2:
    ; k-step 0 complete.  vmcnt retires in issue order: younger than k-step 0's pieces are k-step 1's 16 pieces and the
    ; %[nops] vector-memory operations the previous tile's epilogue issued behind them (a LOWER bound is safe)
    s_cmp_ge_u32 %[nops], 47
    s_cbranch_scc1 5f
    s_cmp_ge_u32 %[nops], 32
    s_cbranch_scc1 6f
    s_waitcnt vmcnt(16)
    s_branch 7f
5:
    s_waitcnt vmcnt(63)
    s_branch 7f
6:
    s_waitcnt vmcnt(48)
7:
    s_barrier
    .irp f,0,1,2,3,4,5,6,7
      ds_read_b128 v[128+4*\f:128+4*\f+3], %[rdW0] offset:128*\f
    .endr
    .irp f,0,1,2,3,4,5,6,7
      ds_read_b128 v[160+4*\f:160+4*\f+3], %[rdX0] offset:128*\f
    .endr
    s_waitcnt lgkmcnt(0)
    ; ---- k-steps 0, 1 (accumulators start from zero in k-step 0)
    NT4W_KSTEP 1, 1, 1, 24, 16, %[rdX0], %[rdW0], %[rdX1], %[rdW1], s50, s52
    NT4W_KSTEP 0, 1, 1, 24, 16, %[rdX1], %[rdW1], %[rdX0], %[rdW0], s51, s53
    s_cmp_eq_u32 s48, 0
    s_cbranch_scc1 4f
3:
    NT4W_KSTEP 0, 1, 1, 24, 16, %[rdX0], %[rdW0], %[rdX1], %[rdW1], s50, s52
    NT4W_KSTEP 0, 1, 1, 24, 16, %[rdX1], %[rdW1], %[rdX0], %[rdW0], s51, s53
    s_sub_u32 s48, s48, 1
    s_cmp_eq_u32 s48, 0
    s_cbranch_scc0 3b
4:
    s_cmp_eq_u32 %[more], 0
    s_cbranch_scc1 8f
    ; ---- last two k-steps: their loads are the NEXT tile's k-steps 0 and 1
    s_mov_b64 s[40:41], %[nbX]
    s_mov_b32 s42, %[nnX]
    s_mov_b64 s[44:45], %[nbW]
    s_mov_b32 s46, %[nnW]
    NT4W_KSTEP 0, 1, 1, 24, 16, %[rdX0], %[rdW0], %[rdX1], %[rdW1], s50, s52
    NT4W_KSTEP 0, 1, 0, 0, 0, %[rdX1], %[rdW1], %[rdX0], %[rdW0], s51, s53
    NT4W_BIAS_DMA %[bgo], %[nbias], %[nbdst], %[lane16]
    s_branch 10f
8:
    ; ---- last tile of this workgroup: nothing to load; k-step nk-1's 16 pieces are the youngest operations (exact waits 8 / 0)
    NT4W_KSTEP 0, 0, 1, 8, 0, %[rdX0], %[rdW0], %[rdX1], %[rdW1], s50, s52
    NT4W_KSTEP 0, 0, 0, 0, 0, %[rdX1], %[rdW1], %[rdX0], %[rdW0], s51, s53
10:
    s_nop 15
  )ASM" NT4W_ASM_PURGE
               :
               : [rdX0] "v"(rdX0), [rdX1] "v"(rdX1), [rdW0] "v"(rdW0), [rdW1] "v"(rdW1), [voX] "v"(voX), [voW] "v"(voW),
                 [cbX] "s"(cur.baseX), [cbW] "s"(cur.baseW), [cnX] "s"(cur.nrX), [cnW] "s"(cur.nrW), [nbX] "s"(nxt.baseX),
                 [nbW] "s"(nxt.baseW), [nnX] "s"(nxt.nrX), [nnW] "s"(nxt.nrW), [lda2] "s"(lda2), [ldb2] "s"(ldb2),
                 [mdst] "s"(mdst), [nblk] "s"(nblk), [first] "s"(first), [more] "s"(more), [nops] "s"(nops), [lane16] "v"(lane16), [bgo] "s"(bgo), [cbias] "s"(cbias), [nbias] "s"(nbias),
                 [cbdst] "s"(cbdst), [nbdst] "s"(nbdst)
               : "memory", "vcc", "scc", NT4W_CLOBBER_A, NT4W_CLOBBER_V, NT4W_CLOBBER_S);
}

template <int N>
__device__ __forceinline__ float nt4w_acc() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "i"(N));
  return x;
}
template <int I, int N, class F>
__device__ __forceinline__ void nt4w_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    nt4w_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned nt4w_sgpr(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long nt4w_sgpr64(unsigned long long x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));  // (the builtin returns int: no sign extension into the high word)
  return ((unsigned long long)hi << 32) | lo;
}

template <int EPI, bool RES, bool CS>
__global__ __launch_bounds__(256) void gemm_nt4w_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = p.N >> 8;
  const int tiles_m = (p.M + 255) >> 8;
  const int tiles = tiles_m * tiles_n;
  const int total = tiles * p.batch;
  const unsigned lds0 = lds_addr_of(dsmem);

  const unsigned mdst = nt4w_sgpr(lds0 + wave * 8 * NT4W_BLK);
  const unsigned lda2 = nt4w_sgpr((unsigned)p.lda * 2u), ldb2 = nt4w_sgpr((unsigned)p.ldb * 2u);
  const unsigned nblk = nt4w_sgpr((unsigned)((p.K >> 6) - 4) >> 1);  // k-step pairs inside the loop (2 before, 2 after)

  auto tile_of = [&](int t, int& bz, int& tm, int& tn) {
    bz = t / tiles;
    band_coords(xcd_remap(t - bz * tiles, tiles), tiles_m, tiles_n, tm, tn, p.band);
  };
  auto desc_of = [&](int t, Nt4wTile& d) {
    if (t < total) {
      int bz, tm, tn;
      tile_of(t, bz, tm, tn);
      const int rows = p.M - (tm << 8) < 256 ? p.M - (tm << 8) : 256;
      d.baseX = nt4w_sgpr64((unsigned long long)(p.A + (long)bz * p.sA + (long)(tm << 8) * p.lda));
      d.baseW = nt4w_sgpr64((unsigned long long)(p.B + (long)bz * p.sB + (long)(tn << 8) * p.ldb));
      d.nrX = nt4w_sgpr((unsigned)rows * lda2);
      d.nrW = nt4w_sgpr(256u * ldb2);
    } else {  // no next tile: never used (the K loop's last two k-steps then load nothing)
      d.baseX = nt4w_sgpr64((unsigned long long)p.A);
      d.baseW = nt4w_sgpr64((unsigned long long)p.B);
      d.nrX = 0;
      d.nrW = 0;
    }
  };

  int t = blockIdx.x;
  if (t >= total) return;
  Nt4wTile cur, nxt;
  desc_of(t, cur);
  unsigned first = 1;
  constexpr bool RD_AUX = (EPI == WFT_EPI_MUL_AUX), WR_AUX = (EPI == WFT_EPI_GELU_GRAD);
  // ONE-BYTE gelu' (round 6): aux is not a [M, N] bf16 matrix but 16 KiB per (tile, wave) in FRAGMENT order — the 16 bytes a lane
  // holds for X fragment fx and W fragment pair up (columns 32 (2 up) + 8 q .. +7 and 32 (2 up + 1) + 8 q .. +7 of row 16 fx + mr) at
  // ((tile * 4 + wave) * 16 + up * 8 + fx) * 1024 + lane * 16.  Every store / load instruction moves one contiguous KiB (8 whole lines);
  // the writer (fc1's forward GEMM) and the reader (fc2's backward-data GEMM) have the same M, N and therefore the same tiles.
  // Code q = round(200 g') + 26: a grid of 1/200 on which gelu' = 0 and gelu' = 1 (saturated units) are exact, range [-0.13, 1.145]
  // (gelu' lies in [-0.129, 1.129]), |error| <= 0.0025.  Halves the second output's bytes on the per-CU memory path that prices these
  // two epilogue variants (profiles/README.md §3) and one byte per MLP activation element of saved-for-backward memory.
  constexpr bool RD_AUX8 = (EPI == WFT_EPI_MUL_AUX8), WR_AUX8 = (EPI == WFT_EPI_GELU_GRAD8);
  constexpr bool has_res = RES;  // (a residual operand; MUL_AUX reads aux instead)
  // vector-memory operations every wave issues per tile AFTER the K loop (a lower bound: the column-sum stores are not counted):
  // 32 stores of C, 32 of aux (GELU_GRAD), 32 loads of aux (MUL_AUX), 32 loads of the residual.  They are buffer operations
  // bounded by the tile's valid rows: no row masks, no branches, so the count holds on ragged tiles too.
  const unsigned nops = nt4w_sgpr(32u + (RD_AUX || WR_AUX ? 32u : 0u) + (RD_AUX8 || WR_AUX8 ? 16u : 0u) + (has_res ? 32u : 0u));
  const unsigned bgo = nt4w_sgpr((p.bias != nullptr && wave == 0) ? 1u : 0u);
  int bpar = 0;  // which bias slice this tile reads
  if (!p.bias) {  // no bias: both slices hold zeros for the whole launch (ordered before their first read by the K loop's barriers)
    *(f32x2*)(dsmem + NT4W_BIAS + tid * 8) = f32x2{0.f, 0.f};
  }

  for (; t < total; t += gridDim.x) {
    int bz, tm, tn;
    tile_of(t, bz, tm, tn);
    const int m0 = tm << 8, n0 = tn << 8;
    desc_of(t + gridDim.x, nxt);

    // The tile's 256 bias values travel like the operands: one LDS-DMA piece issued by wave 0 in the PREVIOUS tile's K loop
    // (its head for a workgroup's first tile) into one of two 1 KiB slices.  A plain load before the loop would hold 32
    // registers across it, one after the loop would queue behind the next tile's LDS-DMA pieces (vmcnt retires in order).
    int nbz, ntm, ntn = 0;
    if (t + (int)gridDim.x < total) tile_of(t + gridDim.x, nbz, ntm, ntn);
    const unsigned long long cbias = nt4w_sgpr64((unsigned long long)(p.bias + n0)), nbias = nt4w_sgpr64((unsigned long long)(p.bias + (ntn << 8)));
    const unsigned cbdst = nt4w_sgpr(lds0 + NT4W_BIAS + bpar * 1024), nbdst = nt4w_sgpr(lds0 + NT4W_BIAS + (bpar ^ 1) * 1024);

    // Lane constants of the K loop are RE-DERIVED per tile from an opaque copy of the thread id: kept live across the epilogue
    // (where 64 operand registers + the accumulator traffic need the room) hipcc spills them to scratch, and the reload in front of
    // the K loop is a vmcnt(0) that drains the previous tile's stores.
    unsigned tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int lane_o = tid_o & 63;
    // fragment read addresses: lane l reads row l & 15 of fragment f (f * 128 in the immediate) at k-chunk l >> 4 (+ 4 for k 32-63)
    const unsigned rdX0 = lds0 + (wm * 16 + (lane_o & 15)) * NT4W_BLK + (lane_o >> 4) * 16;
    const unsigned rdW0 = lds0 + NT4W_OP + (wn * 16 + (lane_o & 15)) * NT4W_BLK + (lane_o >> 4) * 16;
    const unsigned rdX1 = rdX0 + NT4W_BUF, rdW1 = rdW0 + NT4W_BUF;
    // LDS-DMA share of this wave: blocks wave*8 .. +7 of each operand; block (h, r) = rows h*128 + rowmap(f, r), f = lane_o >> 3.
    //   X: row = 16 f + r              (fragment f = 16 consecutive rows of A)
    //   W: row = 32 (f >> 1) + 8 (r >> 2) + 4 (f & 1) + (r & 3): result register j of lane group q in fragments 2u, 2u+1 =
    //      output columns 32 u + 8 q + 4 (f & 1) + j -> 8 consecutive columns per lane and fragment pair
    const int ldf = lane_o >> 3, ldc8 = lane_o & 7;
    const int r0 = (wave & 1) * 8;
    const unsigned voX = (unsigned)(((wm * 128 + 16 * ldf + r0) * (int)p.lda + ldc8 * 8) * 2);
    const unsigned voW = (unsigned)(((wm * 128 + 32 * (ldf >> 1) + 4 * (ldf & 1) + 8 * (r0 >> 2)) * (int)p.ldb + ldc8 * 8) * 2);
    nt4w_kloop(rdX0, rdX1, rdW0, rdW1, voX, voW, cur, nxt, lda2, ldb2, mdst, nblk, first, nt4w_sgpr(t + (int)gridDim.x < total ? 1u : 0u), nops, lane_o * 16, nt4w_sgpr(bgo), cbias, nbias, cbdst, nbdst);
    first = 0;
    cur = nxt;

    // The epilogue's lane coordinates are derived HERE, from an opaque copy of the thread id: computed in front of the K loop they
    // are live across it, and in the variants that need the most registers behind it (column sums) hipcc spilled them — the
    // reload in front of the NEXT K loop is a vmcnt(0) that drains this tile's stores (the seam the persistent loop exists to hide).
    // lane (q, mr) holds, for X fragment fx and W fragment pair u, the 8 consecutive columns n0 + wn*128 + 32 u + 8 q .. +7 of
    // row m0 + wm*128 + 16 fx + mr
    unsigned tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int q = (tid_e & 63) >> 4, mr = tid_e & 15;
    const int ncol = n0 + wn * 128 + 8 * q;
    // (without a bias the slices hold zeros, written once below)
    const float* const bias_l = (const float*)(dsmem + NT4W_BIAS + bpar * 1024) + wn * 128 + 8 * q;
    bpar ^= 1;

    // ---- epilogue straight from the accumulators.  C / residual / aux are addressed through buffer descriptors that end at the
    // tile's last valid row: stores to rows >= M are dropped and loads return zero in hardware (the row offset is in the
    // per-lane offset, which is what the range check sees).
    const int rows = p.M - m0 < 256 ? p.M - m0 : 256;
    unsigned after_loop;  // an opaque 0 defined here: offsets built on it cannot be hoisted above the K loop, where registers are scarce
    asm volatile("v_mov_b32 %0, 0" : "=v"(after_loop));
    const unsigned row_l = (unsigned)(wm * 128 + mr) + after_loop;  // + 16 fx
    auto srd_of = [&](const void* base, long ld) {
      const unsigned long long b = nt4w_sgpr64((unsigned long long)base);
      return __builtin_amdgcn_make_buffer_rsrc((void*)b, (short)0, (int)nt4w_sgpr((unsigned)rows * (unsigned)ld * 2u), 0x00020000);
    };
    // WFT_GEMM_DIAG=22 (timing only): every C store dropped (a descriptor of zero records: the instructions still issue).  Measured
    // at 102000 x 5120 x 1280 with half-line stores: 1 215 TF/s, 1 430 without the stores' traffic, 1 267 with whole lines; handing
    // half of the stores to the next tile's K loop (2 per k-step) costs exactly what it saves — the per-CU memory path, shared with
    // the K loop's 64 KiB per k-step, is the limit (profiles/README.md, round 4).
    const auto srdC = srd_of((const unsigned short*)p.C + (long)bz * p.sC + (long)m0 * p.ldc, (p.diag == 22 || p.diag == 23) ? 0 : p.ldc);
    const unsigned offC = (row_l * (unsigned)p.ldc + (unsigned)ncol) * 2u, stepC = 32u * (unsigned)p.ldc;
    // The 32 groups (fx, u) of a lane are walked in column-chunk PAIRS: order o = 16 up + 2 fx + (u & 1), u = 2 up + (u & 1): both
    // chunks of a row band are converted, then stored together as whole 128-byte lines (store_pair below).
    // Residual / aux rows of this lane: a ring of 16 groups (64 registers; all 32 would leave the lane constants no room beside
    // them): the first 16 in walking order are fetched up front, group o + 16 as soon as o has been consumed.
    u32x4 opq[16];
    const unsigned short* const ob = RD_AUX ? p.aux + (long)bz * p.sAux + (long)m0 * p.ldaux : p.res + (long)bz * p.sR + (long)m0 * p.ldr;
    const long ldo = RD_AUX ? p.ldaux : p.ldr;
    const auto srdO = srd_of((RD_AUX || has_res) ? (const void*)ob : (const void*)p.C, (RD_AUX || has_res) ? ldo : 0);
    const unsigned offO = (row_l * (unsigned)ldo + (unsigned)ncol) * 2u, stepO = 32u * (unsigned)ldo;
    auto load_o = [&](auto oc) {  // group number oc in walking order
      constexpr int o = decltype(oc)::value, fx = (o >> 1) & 7, u = 2 * (o >> 4) + (o & 1);
      opq[o & 15] = __builtin_amdgcn_raw_buffer_load_b128(srdO, offO + fx * stepO + 64u * u, 0, 0);
    };
    if constexpr (RD_AUX || has_res) nt4w_for<0, 16>(load_o);
    const auto srdA = srd_of(WR_AUX ? p.aux + (long)bz * p.sAux + (long)m0 * p.ldaux : (const unsigned short*)p.C, (WR_AUX && p.diag != 23 && p.diag != 24) ? p.ldaux : 0);  // (DIAG 23: C and aux stores dropped, 24: aux only — timing)
    const unsigned stepA = 32u * (unsigned)p.ldaux;  // (wave-uniform by construction: no readfirstlane — it would cost a vector register across the K loop)
    // one-byte gelu': this (tile, wave)'s 16 KiB block; lane offset 16 (lane) + 1024 (up * 8 + fx)
    const unsigned long long a8base = nt4w_sgpr64((unsigned long long)((const char*)p.aux + (((long)bz * tiles + (long)tm * tiles_n + tn) * 4 + wave) * 16384));
    const auto srdA8 = __builtin_amdgcn_make_buffer_rsrc((void*)a8base, (short)0, (RD_AUX8 || WR_AUX8) ? 16384 : 0, 0x00020000);
    const unsigned offA8 = ((tid_e & 63) << 4) + after_loop;
    u32x4 a8q[16];
    if constexpr (RD_AUX8)
      nt4w_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; a8q[i] = __builtin_amdgcn_raw_buffer_load_b128(srdA8, offA8 + 1024u * i, 0, 0); });
    // Whole-line stores.  As they leave the MFMA a lane (q, mr) holds 16 bytes of row mr in each column chunk, so one store
    // instruction would write 16 rows x 64 B: half lines, and the CU's store path is paid per request (measured: whole lines cost
    // 27 % less, profiles/README.md round 4).  Two DPP moves per dword (row_ror:8 = lanes mr <-> mr + 8 of a 16-lane row, bank
    // masks) regroup a chunk pair: store 1 = rows 0-7 of the band, store 2 = rows 8-15, every row a full 128-byte line
    //   lanes mr < 8 : [own chunk u]              | [chunk u of lane mr + 8]
    //   lanes mr >= 8: [chunk u+1 of lane mr - 8] | [own chunk u+1]
    const unsigned lrow = (unsigned)(wm * 128 + (mr & 7)) + after_loop, lcol = (unsigned)(n0 + wn * 128 + 32 * (mr >> 3) + 8 * q);
    const unsigned offCL = (lrow * (unsigned)p.ldc + lcol) * 2u, offAL = (lrow * (unsigned)p.ldaux + lcol) * 2u;
    auto store_pair = [&](auto srd, unsigned off, unsigned half_step, const u32x4& pu, const u32x4& pv) {
      u32x4 s1, s2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] = (unsigned)__builtin_amdgcn_update_dpp((int)pu[e], (int)pv[e], 0x128, 0xf, 0xc, false);
        s2[e] = (unsigned)__builtin_amdgcn_update_dpp((int)pv[e], (int)pu[e], 0x128, 0xf, 0x3, false);
      }
      __builtin_amdgcn_raw_buffer_store_b128(s1, srd, off, 0, NT4W_ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(s2, srd, off + half_step, 0, NT4W_ST_AUX);
    };
    nt4w_for<0, 2>([&](auto upc) {
      constexpr int up = decltype(upc)::value;
      float cs[2][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[0][e] = 0.f; cs[1][e] = 0.f; }
      nt4w_for<0, 8>([&](auto fxc) {
        constexpr int fx = decltype(fxc)::value;
        u32x4 pk[2], dpk[2];
        nt4w_for<0, 2>([&](auto hc) {
          constexpr int hh = decltype(hc)::value, u = 2 * up + hh, o = 16 * up + 2 * fx + hh;
          constexpr int i0 = (fx * 8 + 2 * u) * 4, i1 = i0 + 4;
          const f32x4 b0 = *(const f32x4*)(bias_l + 32 * u), b1 = *(const f32x4*)(bias_l + 32 * u + 4);
          float v[8];
          v[0] = nt4w_acc<i0>(); v[1] = nt4w_acc<i0 + 1>(); v[2] = nt4w_acc<i0 + 2>(); v[3] = nt4w_acc<i0 + 3>();
          v[4] = nt4w_acc<i1>(); v[5] = nt4w_acc<i1 + 1>(); v[6] = nt4w_acc<i1 + 2>(); v[7] = nt4w_acc<i1 + 3>();
          if constexpr (!RD_AUX8) {  // (the one-byte backward-data form has alpha = 1 and no bias by contract: wft_nt4w_eligible)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v[e] * p.alpha + b0[e]; v[4 + e] = v[4 + e] * p.alpha + b1[e]; }
          }
          if constexpr (EPI == WFT_EPI_GELU_GRAD) {
            float dv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gelu_both_f(v[e], v[e], dv[e]);
            dpk[hh] = u32x4{pack2bf(dv[0], dv[1]), pack2bf(dv[2], dv[3]), pack2bf(dv[4], dv[5]), pack2bf(dv[6], dv[7])};
          } else if constexpr (WR_AUX8) {
            // q = round-to-nearest-even(200 g') + 26 through the fp32 magic constant 1.5 * 2^23 (+ 26): the code is the low byte of the
            // sum's bit pattern; four codes are gathered into a dword by two v_perm_b32 and an or
            unsigned qb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float dv;
              gelu_both_f(v[e], v[e], dv);
              qb[e] = __builtin_bit_cast(unsigned, fmaf(dv, 200.0f, 12582912.0f + 26.0f));
            }
            dpk[0][2 * hh] = __builtin_amdgcn_perm(qb[1], qb[0], 0x0c0c0400u) | __builtin_amdgcn_perm(qb[3], qb[2], 0x04000c0cu);
            dpk[0][2 * hh + 1] = __builtin_amdgcn_perm(qb[5], qb[4], 0x0c0c0400u) | __builtin_amdgcn_perm(qb[7], qb[6], 0x04000c0cu);
          } else if constexpr (RD_AUX8) {
            const u32x4 a4 = a8q[up * 8 + fx];
            const unsigned w0 = a4[2 * hh], w1 = a4[2 * hh + 1];  // ((float)((w >> 8 k) & 255) compiles to v_cvt_f32_ubyte<k>)
            // decode + multiply on value PAIRS: two v_cvt_f32_ubyte, one v_pk_fma_f32, one v_pk_mul_f32 per pair
            const f32x2 k5 = f32x2{0.005f, 0.005f}, k13 = f32x2{-0.13f, -0.13f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const unsigned w = e < 4 ? w0 : w1;
              const f32x2 qf = f32x2{(float)((w >> (8 * (e & 3))) & 0xffu), (float)((w >> (8 * (e & 3) + 8)) & 0xffu)};
              const f32x2 gd = __builtin_elementwise_fma(qf, k5, k13);
              const f32x2 pr = f32x2{v[e], v[e + 1]} * gd;
              v[e] = pr[0]; v[e + 1] = pr[1];
            }
          } else if constexpr (EPI == WFT_EPI_MUL_AUX) {
            const u32x4 a4 = opq[o & 15];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] *= __builtin_bit_cast(float, a4[e] << 16);
              v[2 * e + 1] *= __builtin_bit_cast(float, a4[e] & 0xffff0000u);
            }
          }
          if constexpr (!RD_AUX && has_res) {
            const u32x4 r4 = opq[o & 15];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += p.beta * __builtin_bit_cast(float, r4[e] << 16);
              v[2 * e + 1] += p.beta * __builtin_bit_cast(float, r4[e] & 0xffff0000u);
            }
          }
          if constexpr ((RD_AUX || has_res) && o < 16) load_o(std::integral_constant<int, o + 16>{});
          pk[hh] = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          if constexpr (CS) {
            // rows >= M add exact zeros: CS is only instantiated with MUL_AUX (no bias), their accumulators are products of the
            // zeros the A descriptor returns past M, and so are their aux values — no select per element
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[hh][e] += v[e];
          }
        });
        if constexpr (WR_AUX) store_pair(srdA, offAL + fx * stepA + 128u * up, stepA >> 1, dpk[0], dpk[1]);
        if constexpr (WR_AUX8) __builtin_amdgcn_raw_buffer_store_b128(dpk[0], srdA8, offA8 + 1024u * (up * 8 + fx), 0, NT4W_ST_AUX);
        if (p.diag == 30) {  // (A/B: half-line stores, default cache policy — the first form of this epilogue)
          __builtin_amdgcn_raw_buffer_store_b128(pk[0], srdC, offC + fx * stepC + 128u * up, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(pk[1], srdC, offC + fx * stepC + 128u * up + 64u, 0, 0);
        } else {
          store_pair(srdC, offCL + fx * stepC + 128u * up, stepC >> 1, pk[0], pk[1]);
        }
      });
      if constexpr (CS) {
        // column sums of this wave's 128 rows in chunks 2 up, 2 up + 1: 16 values per lane (8 hh + e), summed over the 16 row lanes mr by
        // a halving butterfly — a lane keeps the half of its values its lane bit selects and adds the partner's copy of that half:
        // 15 exchanges instead of 64; lane mr ends with value index mr and stores that one float
        float cv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[i] = cs[i >> 3][i & 7];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
          const bool upper = (mr & m) != 0;
#pragma unroll
          for (int i = 0; i < m; ++i) {
            const float keep = upper ? cv[i + m] : cv[i], send = upper ? cv[i] : cv[i + m];
            cv[i] = keep + __shfl_xor(send, m, 64);
          }
        }
        // (row base formed on the scalar unit: a 64-bit per-lane product would keep p.N in a vector register across the K loop)
        float* const csb = (float*)nt4w_sgpr64((unsigned long long)(p.cs_part + (long)(tm * 2 + wm) * p.N));
        csb[ncol + 32 * (2 * up + (mr >> 3)) + (mr & 7)] = cv[0];
      }
    });
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// Eligibility beyond nt_uses_256 (gemm.hip): bf16 C, the training epilogues (bias / residual / GELU_GRAD / MUL_AUX / column sums;
// GELU with a pre-activation output, DGELU, row-period zeroing and fp32 / accumulating outputs stay on the ping-pong kernel), K a
// multiple of 128 and >= 256 (the k loop is unrolled in k-step pairs around a two-step head and tail), 32-bit offsets inside a
// 256-row tile of every operand.
bool wft_nt4w_eligible(const wft_gemm_args* a) {
  const bool aux8 = a->epilogue == WFT_EPI_GELU_GRAD8 || a->epilogue == WFT_EPI_MUL_AUX8;
  const bool epi_ok = a->epilogue == WFT_EPI_NONE || ((a->epilogue == WFT_EPI_GELU_GRAD || a->epilogue == WFT_EPI_MUL_AUX || aux8) && !a->residual);
  if (a->colsum && a->epilogue != WFT_EPI_MUL_AUX && a->epilogue != WFT_EPI_MUL_AUX8) return false;  // (fused column sums exist for the fc2 backward-data product only)
  if (aux8 && (a->batch != 1 || a->alpha != 1.f)) return false;
  return epi_ok && !a->c_is_f32 && !a->accumulate && a->K % 128 == 0 && a->K >= 256 && a->N % 256 == 0 && a->lda >= a->K &&
         a->valid_rows_period == 0 && !a->residual_first && 256 * a->lda * 2 < (1ll << 31) && 256 * a->ldb * 2 < (1ll << 31) &&
         256 * a->ldc * 2 < (1ll << 31) && 256 * a->ldr * 2 < (1ll << 31) && (aux8 || 256 * a->ldaux * 2 < (1ll << 31));
}

int wft_nt4w_launch(const wft_gemm_args* a, const GemmP& p, bool persistent, void* stream) {
  const long t256 = ((a->M + 255) / 256) * (a->N / 256) * a->batch;
  const int ncu = wft_num_cus();
  dim3 grid((unsigned)((t256 < ncu || !persistent) ? t256 : ncu)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_4W(E, R, CSF)                                              \
  do {                                                                    \
    auto kfn = gemm_nt4w_kernel<E, R, CSF>;                               \
    static DynLdsOnce once;                                               \
    if (!once.set(kfn, NT4W_LDS)) return WFT_ERR_LAUNCH;                                              \
    hipLaunchKernelGGL(kfn, grid, block, NT4W_LDS, s, p);                 \
  } while (0)
  const bool res = a->residual != nullptr, csf = p.cs_part != nullptr;
  switch (a->epilogue) {
    case WFT_EPI_NONE: if (res) LAUNCH_4W(WFT_EPI_NONE, true, false); else LAUNCH_4W(WFT_EPI_NONE, false, false); break;
    case WFT_EPI_GELU_GRAD: LAUNCH_4W(WFT_EPI_GELU_GRAD, false, false); break;
    case WFT_EPI_MUL_AUX: if (csf) LAUNCH_4W(WFT_EPI_MUL_AUX, false, true); else LAUNCH_4W(WFT_EPI_MUL_AUX, false, false); break;
    case WFT_EPI_GELU_GRAD8: LAUNCH_4W(WFT_EPI_GELU_GRAD8, false, false); break;
    case WFT_EPI_MUL_AUX8: if (csf) LAUNCH_4W(WFT_EPI_MUL_AUX8, false, true); else LAUNCH_4W(WFT_EPI_MUL_AUX8, false, false); break;
    default: wft_set_error("wft_gemm_nt_bf16: unknown epilogue %d", a->epilogue); return WFT_ERR_ARG;
  }
#undef LAUNCH_4W
  return WFT_OK;
}
