// attn.hip — fused attention for head_dim 64 (every Whisper size): forward with online
// softmax, and a two-kernel recompute backward (dQ sweep over keys; dK/dV sweep over
// queries) — no atomics, bitwise reproducible.
//
// MFMA plan (v_mfma_f32_32x32x16_bf16, one wave = 32 queries (fwd, dq) or 32 keys (dkdv)):
//   fwd : S^T = K·Q^T (key rows from LDS, Q in registers; the query sits on the LANE, so
//         the row max / row sum of softmax are in-lane reductions + one xor-32 shuffle),
//         O^T += V^T·P^T with P^T taken straight from the S^T accumulators
//         (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand")
//         and V^T fragments read with ds_read_b64_tr_b16.
//   dq  : same orientation; dP^T = V·dO^T, dS^T = P^T ⊙ (dP^T − δ), dQ^T += K^T·dS^T.
//   dkdv: key on the lane: S = Q·K^T, dP = dO·V^T (K, V rows in registers, Q/dO tiles in
//         LDS), dV^T += dO^T·P, dK^T += Q^T·dS.
// LDS tiles are [64 rows][64 bf16] (128-byte rows) filled by global_load_lds_dwordx4 with
// ONE source-side swizzle (chunk ^= F(row)) that is conflict-free for both the 32-row
// ds_read_b128 operand reads and the 4-row transposed reads.
#include "common.h"
#include <stdlib.h>

#ifndef ATT_PK
#define ATT_PK 1
#endif
#define ATT_NEG (-1.0e30f)
#define ATT_TAU 8.0f  // lazy-rescale threshold (log2 units)
#define LOG2E 1.4426950408889634f

struct AttnP {
  const unsigned short* q; long ldq, q_bs;
  const unsigned short* k; long ldk, k_bs;
  const unsigned short* v; long ldv, v_bs;
  unsigned short* o; long ldo, o_bs;
  float* lse;
  int B, H, Tq, Tk, causal;
  float scale;
  const unsigned short* d_o; long lddo, do_bs;
  float* delta;
  unsigned short* dq; long lddq, dq_bs;
  unsigned short* dk; long lddk, dk_bs;
  unsigned short* dv; long lddv, dv_bs;
  float* cs_q;  // [B * ceil(Tq/32)][H*64] per-wave column sums of dq (or NULL)
  float* cs_v;  // [B * ceil(Tk/32)][H*64] per-wave column sums of dv (or NULL)
  int xcd;      // XCD-aware block placement on (WFT_ATTN_XCD=0 switches it off for A/B runs)
};

__device__ __forceinline__ int att_F(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// Stage one [64][64] bf16 tile (rows row0.. of a [nrows, ld] matrix, 64 columns at `base`).
// 8 wave-instructions of 8 rows x 128 B; wave w issues instructions 2w, 2w+1.
// The per-lane part of the source address (row-in-tile * ld + swizzled chunk) is computed ONCE per kernel
// (AttStage); a full tile then costs no vector arithmetic at all: the tile origin is a wave-uniform 64-bit
// base (SALU) and the load uses the saddr + 32-bit-voffset form.  (Before: 16 v_mul_lo_u32 + 8 v_mad_u64_u32
// per tile, ~25 % of the forward kernel's VALU cycles.)  Only the ragged last tile clamps rows per lane.
struct AttStage {
  unsigned off[2];  // byte offset of this lane's 16 bytes inside a tile whose row 0 is the base
  int row[2];
};
__device__ __forceinline__ AttStage att_stage_init(long ld, int wave, int lane) {
  AttStage st;
  const int rr = lane >> 3, cp = lane & 7;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * (wave * 2 + j) + rr;
    st.row[j] = row;
    st.off[j] = (unsigned)(row * (int)ld + ((cp ^ att_F(row)) << 3)) * 2u;
  }
  return st;
}
template <bool RAGGED>
__device__ __forceinline__ void att_stage1(const AttStage& st, const unsigned short* base, long ld, int row0, int nrows,
                                           char* tile, int wave, int lane) {
  const char* tb = (const char*)base + (long)row0 * ld * 2;  // wave-uniform
  if (!RAGGED) {  // scalar base + constant per-lane offset: the saddr form, no vector instruction per piece
    const unsigned long long b64 = (unsigned long long)tb;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b64), hi = __builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr_of(tile) + wave * 2048);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_saddr(st.off[j], sb, dst + j * 1024);
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int rl = st.row[j];
    rl = row0 + rl < nrows ? rl : nrows - 1 - row0;
    const unsigned off = (unsigned)(rl * (int)ld + (((lane & 7) ^ att_F(st.row[j])) << 3)) * 2u;
    glds16(tb + off, tile + (wave * 2 + j) * 1024);
  }
}
__device__ __forceinline__ void att_stage(const AttStage& st, const unsigned short* base, long ld, int row0, int nrows,
                                          char* tile, int wave, int lane) {
  if (row0 + 64 <= nrows) att_stage1<false>(st, base, ld, row0, nrows, tile, wave, lane);
  else att_stage1<true>(st, base, ld, row0, nrows, tile, wave, lane);
}
// two tiles that share row0 / nrows (K and V, or Q and dO): ONE wave-uniform branch for both
__device__ __forceinline__ void att_stage2(const AttStage& sa, const unsigned short* a, long lda, char* ta,
                                           const AttStage& sb, const unsigned short* b, long ldb, char* tb,
                                           int row0, int nrows, int wave, int lane) {
  if (row0 + 64 <= nrows) {
    att_stage1<false>(sa, a, lda, row0, nrows, ta, wave, lane);
    att_stage1<false>(sb, b, ldb, row0, nrows, tb, wave, lane);
  } else {
    att_stage1<true>(sa, a, lda, row0, nrows, ta, wave, lane);
    att_stage1<true>(sb, b, ldb, row0, nrows, tb, wave, lane);
  }
}

// attn.hip is compiled with -ffinite-math-only (Makefile): without it hipcc canonicalises (v_max_f32 x,x,x) every MFMA
// result in front of fmaxf, ~25 extra VALU instructions per tile.  Scores are finite by construction (masked entries
// are -1e30, never -inf).  Plain builtins (not inline asm) so the compiler's MFMA->VALU hazard handling still applies.
__device__ __forceinline__ float att_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float att_max2(float a, float b) { return __builtin_fmaxf(a, b); }
// max over the 32 scores a lane holds for its query (two 32x32 accumulator blocks)
__device__ __forceinline__ float att_max32(const f32x16& a, const f32x16& b) {
  float t0 = att_max3(a[0], a[1], a[2]), t1 = att_max3(a[3], a[4], a[5]);
  float t2 = att_max3(b[0], b[1], b[2]), t3 = att_max3(b[3], b[4], b[5]);
  t0 = att_max3(t0, a[6], a[7]);   t1 = att_max3(t1, a[8], a[9]);
  t2 = att_max3(t2, b[6], b[7]);   t3 = att_max3(t3, b[8], b[9]);
  t0 = att_max3(t0, a[10], a[11]); t1 = att_max3(t1, a[12], a[13]);
  t2 = att_max3(t2, b[10], b[11]); t3 = att_max3(t3, b[12], b[13]);
  t0 = att_max3(t0, a[14], a[15]); t2 = att_max3(t2, b[14], b[15]);
  return att_max3(att_max2(t0, t1), t2, t3);
}

// Per-lane LDS byte offsets of every fragment read, computed ONCE per kernel: with the tile base a
// compile-time constant (loops are unrolled by two over the LDS buffers) every ds_read in the tile loop is
// base-VGPR + immediate — the address arithmetic that used to be ~1/3 of the VALU stream is gone.
//   row[s]      : 32x32x16 A-operand row read, lane (r = lane&31, h = lane>>5) gets tile[blk*32 + r][16s + 8h .. +8]
//                 (+ blk*4096 immediate)
//   tr[db][t]   : transposed read t (rows 8t + 4h + (i>>2)) of the "accumulator as B operand" k-order:
//                 element j of lane (r, h) = tile[16*ks + 8*(j>>2) + 4h + (j&3)][32*db + r]   (+ ks*2048 immediate)
struct AttOffs {
  int row[4];
  int tr[2][2];
};
__device__ __forceinline__ AttOffs att_offsets(int lane) {
  AttOffs o;
  const int r = lane & 31, h = lane >> 5, g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int s = 0; s < 4; ++s) o.row[s] = r * 128 + (((2 * s + h) ^ att_F(r)) << 4);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int rowt = 8 * t + 4 * h + (i >> 2);
      const int col = 32 * db + 16 * (g & 1) + 4 * (i & 3);
      o.tr[db][t] = rowt * 128 + (((col >> 3) ^ att_F(rowt)) << 4) + ((col & 7) << 1);
    }
  return o;
}
__device__ __forceinline__ bf16x8 att_row_frag(const char* tile, const AttOffs& o, int blk, int s) {
  return *(const bf16x8*)(tile + blk * 4096 + o.row[s]);
}
__device__ __forceinline__ bf16x8 att_tr_frag(const char* tile, const AttOffs& o, int ks, int db) {
  s16x8 out;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const s16x4 x = lds_read_tr16(tile + ks * 2048 + o.tr[db][t]);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * t + e] = x[e];
  }
  return __builtin_bit_cast(bf16x8, out);
}

__device__ __forceinline__ bf16x8 att_pack8(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
  return r;
}

__device__ __forceinline__ bf16x8 att_load_reg_frag(const unsigned short* rowptr, int s, int h) {
  return *(const bf16x8*)(rowptr + 16 * s + 8 * h);
}

// ds_read_b64_tr_b16 from inline asm (the caller owns s_waitcnt lgkmcnt + sched_barrier before first use; see
// common.h lds_read_tr16_asm) with the slot / fragment offset in the instruction's immediate field
template <int IMM>
__device__ __forceinline__ s16x4 att_tr_asm(unsigned lds_byte_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}
// ds_read_b128 from inline asm, same contract: hipcc sinks plain LDS loads to just in front of their first use (one LDS round
// trip per MFMA pair in the S / dP loops); issued from asm they stay where they are written — all in one batch.
template <int IMM>
__device__ __forceinline__ f32x4 att_f4_asm(unsigned lds_byte_addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}
template <int IMM>
__device__ __forceinline__ bf16x8 att_row_asm(unsigned lds_byte_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(IMM));
  return r;
}
__device__ __forceinline__ bf16x8 att_join(s16x4 a, s16x4 b) {
  s16x8 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) { out[e] = a[e]; out[4 + e] = b[e]; }
  return __builtin_bit_cast(bf16x8, out);
}
// lanes l and l^32 hold the two halves of one query's row: combine them with v_permlane32_swap (VALU) instead of
// a ds_bpermute round trip through the LDS pipe.  (Inline asm: hipcc folds the builtin's two results into one value
// when both inputs are the same variable.  s_nop 1 covers the VALU-write -> permlane-read hazard.)
__device__ __forceinline__ void att_xhalf(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float att_xhalf_max(float v) {
  float a, b;
  att_xhalf(v, a, b);
  return att_max2(a, b);
}
__device__ __forceinline__ float att_xhalf_sum(float v) {
  float a, b;
  att_xhalf(v, a, b);
  return a + b;
}

// Column sums (over the 32 rows = lanes of one half-wave pair) of a [64 d][32 rows] accumulator pair whose values were
// just rounded to bf16 for the store: acc[db][4a+e] belongs to column d = 32 db + 8 a + 4 h + e.  Rows >= nvalid are
// excluded.  Result: lanes r == 0 (h = 0, 1) write 32 floats each to dst[d].
__device__ __forceinline__ void att_colsum_store(const f32x16 (&acc)[2], float mul, bool row_valid, int r, int h, float* dst) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = row_valid ? bf2f(f2bf(acc[db][i] * mul)) : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);  // xor masks < 32 stay inside the half-wave
      if (r == 0) dst[32 * db + 8 * (i >> 2) + 4 * h + (i & 3)] = v;
    }
}

// XCD-aware block placement.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its
// own 4 MiB L2.  All nx blocks of one (batch, head) re-stream the same K/V (or Q/dO) tiles, so they are mapped onto ONE
// XCD: the i-th workgroup an XCD receives works on group (i / nx) * 8 + xcd, member i % nx.  (PMC before: FETCH_SIZE of
// the backward kernels was ~5x their algorithmic bytes — every XCD pulled every head's tiles through the fabric.)
// Bijective when the number of (batch, head) groups is a multiple of 8; identity order otherwise.
__device__ __forceinline__ void att_block_coords(int nx, int H, int B, int xcd_on, int& bx, int& hd, int& b) {
  const int L = blockIdx.x;
  int g, m;
  if ((((long)H * B) & 7) == 0 && xcd_on) {
    const int xcd = L & 7, i = L >> 3;
    g = (i / nx) * 8 + xcd;
    m = i - (i / nx) * nx;
  } else {
    g = L / nx;
    m = L - g * nx;
  }
  bx = m;
  hd = g % H;
  b = g / H;
}

template <int V>
struct IntC { static constexpr int value = V; };
// f(IntC<0>{}), ..., f(IntC<N-1>{}): loop indices usable as template arguments (immediate offsets of asm reads)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(IntC<N - 1>{});
  }
}

// ------------------------------------------------------------------------------ forward
// K/V tiles travel through a THREE-slot LDS ring, staged two tiles ahead of their use, and the end-of-tile
// wait is a counted s_waitcnt vmcnt(4) (this wave's 4 LDS-DMA instructions of tile kt+2 may stay in flight).
// The transposed V reads are inline asm: with the builtin, hipcc drains every outstanding LDS-DMA
// (s_waitcnt vmcnt(0)) in front of the first ds_read_b64_tr of each tile, which cut the prefetch distance to
// half a tile and left the kernel latency-bound (no-load experiment: +27 %).
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[3 * 16384];  // [slot 3][K 8K | V 8K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: conditions on it are scalar branches, not exec masks
  const int r = lane & 31, h = lane >> 5;
  int bx, hd, b;
  att_block_coords((p.Tq + 127) >> 7, p.H, p.B, p.xcd, bx, hd, b);
  const int q0 = bx * 128;
  const int qw0 = q0 + wave * 32;
  const int qi = qw0 + r;
  const int qc = qi < p.Tq ? qi : p.Tq - 1;
  const unsigned short* qrow = p.q + (long)b * p.q_bs + (long)qc * p.ldq + hd * 64;
  const unsigned short* kb = p.k + (long)b * p.k_bs + hd * 64;
  const unsigned short* vb = p.v + (long)b * p.v_bs + hd * 64;
  const float c = p.scale * LOG2E;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = att_load_reg_frag(qrow, s, h);
  const AttOffs offs = att_offsets(lane);
  const AttStage stK = att_stage_init(p.ldk, wave, lane), stV = att_stage_init(p.ldv, wave, lane);
  const unsigned lds0 = lds_addr_of(smem);
  unsigned tra[2][2];  // absolute LDS byte addresses of the transposed reads in slot 0's K tile
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int t = 0; t < 2; ++t) tra[db][t] = lds0 + offs.tr[db][t];

  int nkt = (p.Tk + 63) >> 6;
  if (p.causal) {
    const int last = (q0 + 127) / 64 + 1;
    nkt = nkt < last ? nkt : last;
  }
  const f32x16 zero16 = f32x16{0};
  f32x16 oacc[2];
  oacc[0] = zero16;
  oacc[1] = zero16;
  // Lazy rescaling: `m` is a STALE running maximum (raw q.k units) that enters the S MFMA chains as their initial
  // accumulator (minit = -m in every register: S' = S - m costs no VALU), and is only raised when a tile's maximum exceeds
  // it by more than ATT_TAU in log2 units (P <= 2^ATT_TAU is exact in bf16's exponent range; l is fp32).  Most tiles then
  // skip the subtraction, the O rescale and the alpha exponential: the kernel is VALU-issue-bound (v_exp_f32 8 cycles,
  // everything else 4).  Same sums as the eager form up to fp32 rounding of l and O.
  float m = 0.f, l = 0.f;
  f32x16 minit = zero16;

  att_stage2(stK, kb, p.ldk, smem, stV, vb, p.ldv, smem + 8192, 0, p.Tk, wave, lane);
  if (nkt > 1) {
    att_stage2(stK, kb, p.ldk, smem + 16384, stV, vb, p.ldv, smem + 16384 + 8192, 64, p.Tk, wave, lane);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 kf[2][4];  // K row fragments of the CURRENT tile; refilled for the next tile behind the mid-tile barrier
#pragma unroll
  for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem, offs, kb2, s);

  // One tile = [stage kt+2 | V^T reads | S MFMAs | softmax | wait + barrier | K reads of kt+1 | PV MFMAs]: every LDS
  // read is issued a phase ahead of its use, and the only barrier sits where tile kt+1 must have landed.
  auto tile = [&](auto cur_tag, int kt) {
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int NXT = (CUR + 1) % 3, NXT2 = (CUR + 2) % 3;
    const int key0 = kt * 64;
    const bool more = kt + 2 < nkt;
    if (more)
      att_stage2(stK, kb, p.ldk, smem + NXT2 * 16384, stV, vb, p.ldv, smem + NXT2 * 16384 + 8192, key0 + 128, p.Tk, wave, lane);
    // (a wave whose 32 queries all lie past the sequence end — T = 1500: the fourth wave of the last 128-query block — only
    // stages and keeps the barriers)
    const bool active = qw0 < p.Tq && !(p.causal && key0 > qw0 + 31);
    s16x4 vt[4][2][2];
    bf16x8 pf[4];
    if (active) {
      static_for<4>([&](auto ks_tag) {  // the k-step's 2048-byte stride rides in the immediate: no address add per read
        constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int t = 0; t < 2; ++t) vt[ks][db][t] = att_tr_asm<CUR * 16384 + 8192 + ks * 2048>(tra[db][t]);
      });
      f32x16 sacc[2];
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][0], qf[0], minit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][s], qf[s], sacc[kb2], 0, 0, 0);
      }
      // mask (only tiles that touch the ragged end / the causal diagonal: wave-uniform branch, selects inside)
      if ((key0 + 64 > p.Tk) || (p.causal && key0 + 63 > qw0)) {
        const int lim = (p.causal ? (qi + 1 < p.Tk ? qi + 1 : p.Tk) : p.Tk) - key0 - 4 * h;  // valid iff key offset < lim
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            sacc[kb2][e] = (32 * kb2 + (e & 3) + 8 * (e >> 2)) < lim ? sacc[kb2][e] : ATT_NEG;
      }
      const float tmax = att_xhalf_max(att_max32(sacc[0], sacc[1]));  // max over the tile of S - m
      if (kt == 0 || __builtin_amdgcn_ballot_w64(tmax * c > ATT_TAU) != 0) {
        // rare path: move the reference maximum (first tile: to the tile's own maximum, whatever its sign)
        const float d = kt == 0 ? tmax : fmaxf(tmax, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d * c);
        m += d;
        l *= alpha;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int e = 0; e < 16; ++e) sacc[kb2][e] -= d;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) minit[e] = -m;
      }
      float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sacc[kb2][e] * c);
          const float p1 = __builtin_amdgcn_exp2f(sacc[kb2][e + 1] * c);
          ls0 += p0;
          ls1 += p1;
          sacc[kb2][e] = p0;
          sacc[kb2][e + 1] = p1;
        }
      l += att_xhalf_sum(ls0 + ls1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pf[ks] = att_pack8(sacc[ks >> 1], ks & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem + NXT * 16384, offs, kb2, s);
    }
    if (active) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_join(vt[ks][db][0], vt[ks][db][1]), pf[ks], oacc[db], 0, 0, 0);
    }
  };
  int kt = 0;
  for (; kt + 2 < nkt; kt += 3) {
    tile(IntC<0>{}, kt);
    tile(IntC<1>{}, kt + 1);
    tile(IntC<2>{}, kt + 2);
  }
  if (kt < nkt) tile(IntC<0>{}, kt);
  if (kt + 1 < nkt) tile(IntC<1>{}, kt + 1);

  if (qi < p.Tq) {
    const float inv = 1.0f / l;
    unsigned short* orow = p.o + (long)b * p.o_bs + (long)qi * p.ldo + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int d = 32 * db + 8 * a + 4 * h;
        u32x2 pk = {pack2bf(oacc[db][4 * a] * inv, oacc[db][4 * a + 1] * inv),
                    pack2bf(oacc[db][4 * a + 2] * inv, oacc[db][4 * a + 3] * inv)};
        *(u32x2*)(orow + d) = pk;
      }
    if (h == 0 && p.lse) p.lse[((long)b * p.H + hd) * p.Tq + qi] = m * p.scale + __logf(l);
  }
}

// ------------------------------------------------------------------------------ delta
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d] and lse enter both backward kernels NEGATED, as the initial accumulators of the
// dP and S MFMA chains (S' = S - lse / scale, so exp2(c S') = exp(scale S - lse) needs no subtraction; dP' = dP - delta).
// Round 3: there is no delta kernel any more.  A lane of the dQ kernel already holds half of its query's dO row for the dP
// product; it loads the same half of the O row, forms its 32 products, adds its partner lane's (the other half: lane ^ 32) and
// has -delta; it writes both constants to the workspace for the dK/dV kernel, which runs behind it on the stream (96 launches and
// a second pass over O and dO per step less: 3.6 ms at 68 clips).

// ------------------------------------------------------------------------------ dQ
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[32768];  // [buf 2][K 8K | V 8K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: conditions on it are scalar branches, not exec masks
  const int r = lane & 31, h = lane >> 5;
  int bx, hd, b;
  att_block_coords((p.Tq + 127) >> 7, p.H, p.B, p.xcd, bx, hd, b);
  const int q0 = bx * 128;
  const int qw0 = q0 + wave * 32;
  const int qi = qw0 + r;
  const int qc = qi < p.Tq ? qi : p.Tq - 1;
  const unsigned short* qrow = p.q + (long)b * p.q_bs + (long)qc * p.ldq + hd * 64;
  const unsigned short* dorow = p.d_o + (long)b * p.do_bs + (long)qc * p.lddo + hd * 64;
  const unsigned short* kb = p.k + (long)b * p.k_bs + hd * 64;
  const unsigned short* vb = p.v + (long)b * p.v_bs + hd * 64;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = att_load_reg_frag(qrow, s, h);
    dof[s] = att_load_reg_frag(dorow, s, h);
  }
  const AttOffs offs = att_offsets(lane);
  const AttStage stK = att_stage_init(p.ldk, wave, lane), stV = att_stage_init(p.ldv, wave, lane);
  const long sidx = ((long)b * p.H + hd) * p.Tq + qc;
  // row constants of this lane's query, negated: the initial accumulators of the S and dP chains
  const float nlse = -p.lse[sidx] / p.scale;
  float ndlt;
  {
    const unsigned short* orow = p.o + (long)b * p.o_bs + (long)qc * p.ldo + hd * 64;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 of = att_load_reg_frag(orow, s, h);
      const u32x4 ou = __builtin_bit_cast(u32x4, of), du = __builtin_bit_cast(u32x4, dof[s]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf2f((unsigned short)(ou[e] & 0xffff)) * bf2f((unsigned short)(du[e] & 0xffff));
        part += bf2f((unsigned short)(ou[e] >> 16)) * bf2f((unsigned short)(du[e] >> 16));
      }
    }
    ndlt = -(part + __shfl_xor(part, 32, 64));  // (a + b == b + a: both lanes of a query hold the same bits)
  }
  if (h == 0 && qi < p.Tq) {  // for the dK/dV kernel
    p.delta[sidx] = ndlt;
    p.delta[(long)p.B * p.H * p.Tq + sidx] = nlse;
  }

  int nkt = (p.Tk + 63) >> 6;
  if (p.causal) {
    const int last = (q0 + 127) / 64 + 1;
    nkt = nkt < last ? nkt : last;
  }
  const float c = p.scale * LOG2E;
  const f32x16 zero16 = f32x16{0};
  f32x16 dqacc[2];
  dqacc[0] = zero16;
  dqacc[1] = zero16;
  f32x16 sinit, pinit;
#pragma unroll
  for (int e = 0; e < 16; ++e) { sinit[e] = nlse; pinit[e] = ndlt; }

  att_stage2(stK, kb, p.ldk, smem, stV, vb, p.ldv, smem + 8192, 0, p.Tk, wave, lane);
  __syncthreads();

  auto tile = [&](auto cur_tag, int kt) {
    constexpr int CUR = decltype(cur_tag)::value;
    const int key0 = kt * 64;
    if (kt + 1 < nkt) {
      att_stage2(stK, kb, p.ldk, smem + (CUR ^ 1) * 16384, stV, vb, p.ldv, smem + (CUR ^ 1) * 16384 + 8192, key0 + 64, p.Tk,
                 wave, lane);
    }
    const char* kt_l = smem + CUR * 16384;
    const char* vt_l = kt_l + 8192;
    if (qw0 < p.Tq && !(p.causal && key0 > qw0 + 31)) {
      f32x16 sacc[2], pacc[2];
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_row_frag(kt_l, offs, kb2, 0), qf[0], sinit, 0, 0, 0);
        pacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_row_frag(vt_l, offs, kb2, 0), dof[0], pinit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) {
          sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_row_frag(kt_l, offs, kb2, s), qf[s], sacc[kb2], 0, 0, 0);
          pacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_row_frag(vt_l, offs, kb2, s), dof[s], pacc[kb2], 0, 0, 0);
        }
      }
      if ((key0 + 64 > p.Tk) || (p.causal && key0 + 63 > qw0)) {
        const int lim = (p.causal ? (qi + 1 < p.Tk ? qi + 1 : p.Tk) : p.Tk) - key0 - 4 * h;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool ok = (32 * kb2 + (e & 3) + 8 * (e >> 2)) < lim;
            const float pv = ok ? __builtin_amdgcn_exp2f(sacc[kb2][e] * c) : 0.f;
            sacc[kb2][e] = pv * pacc[kb2][e];
          }
      } else {
        // packed fp32 (v_pk_mul): two elements per VALU issue slot; the subtractions of lse and delta happened in the MFMAs
        const f32x2 c2 = {c, c};
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            f32x2 t2 = {sacc[kb2][2 * e], sacc[kb2][2 * e + 1]};
            t2 = t2 * c2;
            const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
            f32x2 g2 = {pacc[kb2][2 * e], pacc[kb2][2 * e + 1]};
            g2 = g2 * p2;  // dS^T (unscaled)
            sacc[kb2][2 * e] = g2[0];
            sacc[kb2][2 * e + 1] = g2[1];
          }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 dsf = att_pack8(sacc[ks >> 1], ks & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_tr_frag(kt_l, offs, ks, db), dsf, dqacc[db], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    tile(IntC<0>{}, kt);
    tile(IntC<1>{}, kt + 1);
  }
  if (kt < nkt) tile(IntC<0>{}, kt);

  if (qi < p.Tq) {
    unsigned short* drow = p.dq + (long)b * p.dq_bs + (long)qi * p.lddq + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int d = 32 * db + 8 * a + 4 * h;
        u32x2 pk = {pack2bf(dqacc[db][4 * a] * p.scale, dqacc[db][4 * a + 1] * p.scale),
                    pack2bf(dqacc[db][4 * a + 2] * p.scale, dqacc[db][4 * a + 3] * p.scale)};
        *(u32x2*)(drow + d) = pk;
      }
  }
  if (p.cs_q && qw0 < p.Tq)  // q-projection bias gradient: this wave's 32 queries, summed per head column
    att_colsum_store(dqacc, p.scale, qi < p.Tq, r, h,
                     p.cs_q + ((long)b * ((p.Tq + 31) >> 5) + (qw0 >> 5)) * (p.H * 64) + hd * 64);
}

// ------------------------------------------------------------------------------ dK, dV
// Per 64-query tile the block stages Q, dO (2 x 8 KiB) AND the tile's 64 lse2 / 64 delta values by LDS-DMA
// (global_load_lds_dword: no VGPR round trip, so no compiler-inserted vmcnt(0) in front of an LDS store — that wait used
// to drain the prefetch of the next tile at the START of every tile).  Transposed Q^T / dO^T reads are inline asm
// issued ahead of the S / dP MFMAs of their 32-query half (hipcc drains all LDS-DMA before a ds_read_tr builtin).
// Query tile of the dK/dV sweep: DKDV_Q queries per stage and barrier.  128 (round 3): half the __syncthreads and half the
// LDS-DMA issue phases per MFMA of the 64-query tiles (in-kernel stamps of round 1: 435 + 325 of 4 819 cycles per 64-query tile).
#ifndef DKDV_Q
#define DKDV_Q 128
#endif
#define DKDV_BUF (2 * DKDV_Q * 128 + 2 * DKDV_Q * 4)  // Q [DKDV_Q][64] bf16 | dO | -lse/scale f32 [DKDV_Q] | -delta
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const WFT_GLB void*)gsrc, (WFT_LDS void*)lds_wave_base, 4, 0, 0);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf 2][Q | dO | -lse/scale | -delta], 2 * DKDV_BUF bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: conditions on it are scalar branches, not exec masks
  const int r = lane & 31, h = lane >> 5;
  int bx, hd, b;
  att_block_coords((p.Tk + 127) >> 7, p.H, p.B, p.xcd, bx, hd, b);
  const int k0 = bx * 128;
  const int kw0 = k0 + wave * 32;
  const int ki = kw0 + r;
  const int kc = ki < p.Tk ? ki : p.Tk - 1;
  const unsigned short* krow = p.k + (long)b * p.k_bs + (long)kc * p.ldk + hd * 64;
  const unsigned short* vrow = p.v + (long)b * p.v_bs + (long)kc * p.ldv + hd * 64;
  const unsigned short* qb = p.q + (long)b * p.q_bs + hd * 64;
  const unsigned short* dob = p.d_o + (long)b * p.do_bs + hd * 64;
  const long sbase = ((long)b * p.H + hd) * p.Tq;
  const float* dlt_b = p.delta + sbase;
  const float* lse_b = p.delta + (long)p.B * p.H * p.Tq + sbase;  // -lse / scale, written (like -delta) by the dQ kernel, which runs first
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = att_load_reg_frag(krow, s, h);
    vf[s] = att_load_reg_frag(vrow, s, h);
  }
  const AttOffs offs = att_offsets(lane);
  const AttStage stQ = att_stage_init(p.ldq, wave, lane), stDO = att_stage_init(p.lddo, wave, lane);
  const unsigned lds0 = lds_addr_of(smem);
  // one base register set per buffer (the offsets inside a buffer ride in the instructions' 16-bit immediates)
  unsigned tra[2][2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t = 0; t < 2; ++t) tra[cb][db][t] = lds0 + cb * DKDV_BUF + offs.tr[db][t];
  unsigned rowa[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int s = 0; s < 4; ++s) rowa[cb][s] = lds0 + cb * DKDV_BUF + offs.row[s];
  const unsigned rca = lds0 + 16 * h;  // this lane's first row constant: query 4 h of a 32-query half (+ buffer, + 32 a via immediates)
  const float c = p.scale * LOG2E;
  const f32x2 c2 = {c, c};
  const int nqt = (p.Tq + DKDV_Q - 1) / DKDV_Q;
  const int qt0 = p.causal ? (k0 / DKDV_Q) : 0;  // first query tile that can see key k0
  const f32x16 zero16 = f32x16{0};
  f32x16 dkacc[2], dvacc[2];
  dkacc[0] = zero16; dkacc[1] = zero16;
  dvacc[0] = zero16; dvacc[1] = zero16;

  auto stage_q = [&](char* base, int qt) {
#pragma unroll
    for (int cq = 0; cq < DKDV_Q / 64; ++cq)
      if (qt * DKDV_Q + cq * 64 < p.Tq)  // (a 64-row chunk wholly past the sequence end is neither staged nor read)
        att_stage2(stQ, qb, p.ldq, base + cq * 8192, stDO, dob, p.lddo, base + DKDV_Q * 128 + cq * 8192, qt * DKDV_Q + cq * 64, p.Tq, wave, lane);
    // row constants: wave w stages 64 values — even waves -lse/scale, odd waves -delta, of queries 64 (w >> 1) .. (rows clamped;
    // out-of-range rows are masked later)
    if (wave < DKDV_Q / 32) {
      int qq = qt * DKDV_Q + (wave >> 1) * 64 + lane;
      qq = qq < p.Tq ? qq : p.Tq - 1;
      glds4(((wave & 1) == 0 ? lse_b : dlt_b) + qq, base + 2 * DKDV_Q * 128 + (wave & 1) * (DKDV_Q * 4) + (wave >> 1) * 256);
    }
  };

  if (qt0 < nqt) {
    stage_q(smem, qt0);
    __syncthreads();
  }
  auto tile = [&](auto cur_tag, int qt) {
    constexpr int CUR = decltype(cur_tag)::value;
    const int qq0 = qt * DKDV_Q;
    if (qt + 1 < nqt) stage_q(smem + (CUR ^ 1) * DKDV_BUF, qt + 1);
    if (kw0 < p.Tk && !(p.causal && kw0 > qq0 + DKDV_Q - 1)) {  // (a wave whose 32 keys lie past the end only stages)
      static_for<DKDV_Q / 32>([&](auto qb_tag) {
        constexpr int QB2 = decltype(qb_tag)::value;
        constexpr int qb2 = QB2;
        // a 32-query half that lies entirely past the sequence end or above the causal diagonal contributes nothing
        if (qq0 + 32 * qb2 >= p.Tq || (p.causal && kw0 > qq0 + 32 * qb2 + 31)) return;
        // transposed fragments of this 32-query half: in flight under the S / dP MFMAs and the exponentials
        s16x4 dot[2][2][2], qt_[2][2][2];
        static_for<2>([&](auto ks_tag) {
          constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              dot[ks][db][t] = att_tr_asm<DKDV_Q * 128 + (2 * QB2 + ks) * 2048>(tra[CUR][db][t]);
              qt_[ks][db][t] = att_tr_asm<(2 * QB2 + ks) * 2048>(tra[CUR][db][t]);
            }
        });
        // all eight Q / dO row fragments of the half in one batch behind the transposed reads: ONE LDS round trip in front of
        // the S / dP MFMAs instead of one per k-step (the reads used to be issued pairwise, each pair waited for on the spot)
        bf16x8 aq[4], ad[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          aq[s] = att_row_asm<QB2 * 4096>(rowa[CUR][s]);
          ad[s] = att_row_asm<DKDV_Q * 128 + QB2 * 4096>(rowa[CUR][s]);
        }
        // the half's row constants -lse / scale and -delta: the 16 values a lane needs (queries 8 a + 4 h + e) are laid out
        // exactly like the f32x16 C operand, so they ARE the initial accumulators of the S and dP chains (no VALU at all)
        f32x4 l4[4], d4[4];
        static_for<4>([&](auto a_tag) {
          constexpr int a = decltype(a_tag)::value;
          l4[a] = att_f4_asm<2 * DKDV_Q * 128 + 128 * QB2 + 32 * a>(rca + CUR * DKDV_BUF);
          d4[a] = att_f4_asm<2 * DKDV_Q * 128 + DKDV_Q * 4 + 128 * QB2 + 32 * a>(rca + CUR * DKDV_BUF);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sinit, pinit;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sinit[4 * a + e] = l4[a][e]; pinit[4 * a + e] = d4[a][e]; }
        f32x16 sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], kf[0], sinit, 0, 0, 0);
        f32x16 pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[0], vf[0], pinit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[s], kf[s], sacc, 0, 0, 0);
          pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[s], vf[s], pacc, 0, 0, 0);
        }
        f32x16 dsacc;
        // one decision per 32-query half (wave-uniform): the unmasked body is a single basic block — its eight lse / delta
        // reads, 32 exponentials and the packed arithmetic can be scheduled against each other
        const bool need_mask = (qq0 + 32 * qb2 + 32 > p.Tq) || (kw0 + 32 > p.Tk) || (p.causal && kw0 + 31 > qq0 + 32 * qb2);
        if (need_mask) {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int qg = qq0 + 32 * qb2 + 8 * a + 4 * h + e;
              const bool ok = qg < p.Tq && ki < p.Tk && !(p.causal && ki > qg);
              const float pv = ok ? __builtin_amdgcn_exp2f(sacc[4 * a + e] * c) : 0.f;
              sacc[4 * a + e] = pv;
              dsacc[4 * a + e] = ok ? pv * pacc[4 * a + e] : 0.f;
            }
        } else {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {  // packed fp32 pairs
              f32x2 t2 = {sacc[4 * a + e], sacc[4 * a + e + 1]};
              t2 = t2 * c2;
              const f32x2 p2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
              f32x2 g2 = {pacc[4 * a + e], pacc[4 * a + e + 1]};
              g2 = g2 * p2;
              sacc[4 * a + e] = p2[0];
              sacc[4 * a + e + 1] = p2[1];
              dsacc[4 * a + e] = g2[0];
              dsacc[4 * a + e + 1] = g2[1];
            }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          pf[ks] = att_pack8(sacc, ks);
          dsf[ks] = att_pack8(dsacc, ks);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_join(dot[ks][db][0], dot[ks][db][1]), pf[ks], dvacc[db], 0, 0, 0);
            dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_join(qt_[ks][db][0], qt_[ks][db][1]), dsf[ks], dkacc[db], 0, 0, 0);
          }
      });
    }
    __syncthreads();
  };
  int qt = qt0;
  for (; qt + 1 < nqt; qt += 2) {
    tile(IntC<0>{}, qt);
    tile(IntC<1>{}, qt + 1);
  }
  if (qt < nqt) tile(IntC<0>{}, qt);

  if (ki < p.Tk) {
    unsigned short* dkrow = p.dk + (long)b * p.dk_bs + (long)ki * p.lddk + hd * 64;
    unsigned short* dvrow = p.dv + (long)b * p.dv_bs + (long)ki * p.lddv + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int d = 32 * db + 8 * a + 4 * h;
        u32x2 pk = {pack2bf(dkacc[db][4 * a] * p.scale, dkacc[db][4 * a + 1] * p.scale),
                    pack2bf(dkacc[db][4 * a + 2] * p.scale, dkacc[db][4 * a + 3] * p.scale)};
        *(u32x2*)(dkrow + d) = pk;
        u32x2 pv = {pack2bf(dvacc[db][4 * a], dvacc[db][4 * a + 1]),
                    pack2bf(dvacc[db][4 * a + 2], dvacc[db][4 * a + 3])};
        *(u32x2*)(dvrow + d) = pv;
      }
  }
  if (p.cs_v && kw0 < p.Tk)  // v-projection bias gradient
    att_colsum_store(dvacc, 1.0f, ki < p.Tk, r, h, p.cs_v + ((long)b * ((p.Tk + 31) >> 5) + (kw0 >> 5)) * (p.H * 64) + hd * 64);
}

// ------------------------------------------------------------------------------ host
static int attn_fill(const wft_attn_args* a, AttnP& p) {
  p.q = a->q; p.ldq = a->ldq; p.q_bs = a->q_bs;
  p.k = a->k; p.ldk = a->ldk; p.k_bs = a->k_bs;
  p.v = a->v; p.ldv = a->ldv; p.v_bs = a->v_bs;
  p.o = a->o; p.ldo = a->ldo; p.o_bs = a->o_bs;
  p.lse = a->lse;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = a->causal;
  p.scale = a->scale;
  p.d_o = a->d_o; p.lddo = a->lddo; p.do_bs = a->do_bs;
  p.delta = a->delta;
  p.dq = a->dq; p.lddq = a->lddq; p.dq_bs = a->dq_bs;
  p.dk = a->dk; p.lddk = a->lddk; p.dk_bs = a->dk_bs;
  p.dv = a->dv; p.lddv = a->lddv; p.dv_bs = a->dv_bs;
  p.cs_q = nullptr; p.cs_v = nullptr;
  static const int xcd_on = [] { const char* e = getenv("WFT_ATTN_XCD"); return e ? atoi(e) : 1; }();
  p.xcd = xcd_on;
  return 0;
}

#define ATT_ALIGNED(ptr, ld, bs) ((((uintptr_t)(ptr)) & 15) == 0 && ((ld) % 8) == 0 && ((bs) % 8) == 0)

extern "C" int wft_attn_fwd_bf16(const wft_attn_args* a, void* stream) {
  WFT_CHECK_ARG(a && a->q && a->k && a->v && a->o && a->lse, "null pointer");
  WFT_CHECK_ARG(a->B >= 1 && a->H >= 1 && a->Tq >= 1 && a->Tk >= 1, "bad shape");
  WFT_CHECK_ARG(ATT_ALIGNED(a->q, a->ldq, a->q_bs) && ATT_ALIGNED(a->k, a->ldk, a->k_bs) &&
                    ATT_ALIGNED(a->v, a->ldv, a->v_bs) && ATT_ALIGNED(a->o, a->ldo, a->o_bs),
                "q/k/v/o need 16-byte aligned bases and strides that are multiples of 8");
  WFT_CHECK_ARG(!a->causal || a->Tq == a->Tk, "causal attention needs Tq == Tk");
  AttnP p;
  attn_fill(a, p);
  dim3 grid((unsigned)(((a->Tq + 127) / 128) * a->H * a->B)), block(256);  // 1-D: see att_block_coords
  hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, (hipStream_t)stream, p);
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}

// out[chunk][col] = sum over this chunk's partial rows (fixed order).  64 columns per workgroup (256-byte row segments),
// 4 waves stride the rows; gridDim.y row chunks.  Run twice: [nrows] -> [ATT_CS_CHUNKS] -> [1].
#define ATT_CS_CHUNKS 32
__global__ __launch_bounds__(256) void attn_colsum_reduce_kernel(const float* partial0, long nrows0, float* out0,
                                                                 const float* partial1, long nrows1, float* out1, int n) {
  // blockIdx.z = 0: the q-projection's sums, 1: the v-projection's (one launch per level for both)
  const float* partial = blockIdx.z ? partial1 : partial0;
  const long nrows = blockIdx.z ? nrows1 : nrows0;
  float* out = blockIdx.z ? out1 : out0;
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const long per = (nrows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * per;
  const long r1 = r0 + per < nrows ? r0 + per : nrows;
  float sacc = 0.f;
  if (col < n)
    for (long rr = r0 + wv; rr < r1; rr += 4) sacc += partial[rr * n + col];
  red[wv][cx] = sacc;
  __syncthreads();
  if (wv == 0 && col < n) out[(long)blockIdx.y * n + col] = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
}

extern "C" int64_t wft_attn_bwd_colsum_workspace_bytes(const wft_attn_args* a) {
  if (!a) return 0;
  return ((int64_t)a->B * ((a->Tq + 31) / 32) + (int64_t)a->B * ((a->Tk + 31) / 32) + 2 * ATT_CS_CHUNKS) * a->H * 64 * (int64_t)sizeof(float);
}

extern "C" int wft_attn_bwd_bf16(const wft_attn_args* a, void* stream) {
  WFT_CHECK_ARG(a && a->q && a->k && a->v && a->o && a->lse && a->d_o && a->delta && a->dq && a->dk && a->dv,
                "null pointer");
  WFT_CHECK_ARG(a->B >= 1 && a->H >= 1 && a->Tq >= 1 && a->Tk >= 1, "bad shape");
  WFT_CHECK_ARG(ATT_ALIGNED(a->q, a->ldq, a->q_bs) && ATT_ALIGNED(a->k, a->ldk, a->k_bs) &&
                    ATT_ALIGNED(a->v, a->ldv, a->v_bs) && ATT_ALIGNED(a->o, a->ldo, a->o_bs) &&
                    ATT_ALIGNED(a->d_o, a->lddo, a->do_bs) && ATT_ALIGNED(a->dq, a->lddq, a->dq_bs) &&
                    ATT_ALIGNED(a->dk, a->lddk, a->dk_bs) && ATT_ALIGNED(a->dv, a->lddv, a->dv_bs),
                "tensors need 16-byte aligned bases and strides that are multiples of 8");
  WFT_CHECK_ARG(!a->causal || a->Tq == a->Tk, "causal attention needs Tq == Tk");
  WFT_CHECK_ARG(a->scale > 0.f, "scale must be positive (the row constants are -lse / scale)");
  WFT_CHECK_ARG((!a->dq_colsum && !a->dv_colsum && !a->colsum_ws) || (a->dq_colsum && a->dv_colsum && a->colsum_ws),
                "dq_colsum, dv_colsum and colsum_ws go together");
  AttnP p;
  attn_fill(a, p);
  if (a->colsum_ws) {
    p.cs_q = a->colsum_ws;
    p.cs_v = a->colsum_ws + (long)a->B * ((a->Tq + 31) / 32) * a->H * 64;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(((a->Tq + 127) / 128) * a->H * a->B)), dim3(256), 0, s, p);
  {
    static bool lds_set[64] = {false};  // hipFuncSetAttribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!lds_set[dev]) {
      const hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DKDV_BUF);
      if (e != hipSuccess) {  // (not remembered: the next call tries again)
        wft_set_error("wft_attn_bwd_bf16: the dK/dV kernel needs %d bytes of dynamic LDS (160 KiB per CU: gfx950), hipFuncSetAttribute: %s",
                      2 * DKDV_BUF, hipGetErrorString(e));
        return WFT_ERR_LAUNCH;
      }
      lds_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((unsigned)(((a->Tk + 127) / 128) * a->H * a->B)), dim3(256), 2 * DKDV_BUF, s, p);
  if (p.cs_q) {
    const int n = a->H * 64;
    const long rq = (long)a->B * ((a->Tq + 31) / 32), rk = (long)a->B * ((a->Tk + 31) / 32);
    float* mid_q = p.cs_v + rk * n;
    float* mid_v = mid_q + (long)ATT_CS_CHUNKS * n;
    const dim3 g1((n + 63) / 64, ATT_CS_CHUNKS, 2), g2((n + 63) / 64, 1, 2);
    hipLaunchKernelGGL(attn_colsum_reduce_kernel, g1, dim3(256), 0, s, (const float*)p.cs_q, rq, mid_q, (const float*)p.cs_v, rk, mid_v, n);
    hipLaunchKernelGGL(attn_colsum_reduce_kernel, g2, dim3(256), 0, s, (const float*)mid_q, (long)ATT_CS_CHUNKS, a->dq_colsum,
                       (const float*)mid_v, (long)ATT_CS_CHUNKS, a->dv_colsum, n);
  }
  WFT_CHECK_LAUNCH();
  return WFT_OK;
}
