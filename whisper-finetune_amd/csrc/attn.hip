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
#include "gemm_common.h"  // wft_num_cus
#include <stdlib.h>
#include <string.h>

#ifndef ATT_PK
#define ATT_PK 1
#endif
#define ATT_NEG (-1.0e30f)
#define ATT_TAU 8.0f  // lazy-rescale threshold (log2 units)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

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
  // q_prescaled (wft.h): q already carries scale * log2(e) (folded into the forward weight shadow of the q projection in fp32, one
  // bf16 rounding), so the scores ARE the exponent of exp2 and no kernel multiplies them by c.  c: factor between the q.k
  // accumulators and log2 units (1 when prescaled); ls: factor between them and natural-log units (lse = m * ls + log l; the
  // row constant that enters the S chains of the backward kernels is -lse / ls; dK = dS^T q * ls).  dQ keeps `scale`: the kernels
  // return the gradient w.r.t. the UNSCALED projection output, which is what the projection's backward GEMMs consume.
  int qpre;
  float c, ls;
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

// FWD_ABL=n (developer builds, tools/dev/fwd_abl.sh): attn_fwd_kernel with one ingredient REMOVED — results are wrong, only the
// kernel's duration means anything.  1 no exponentials (P = c S), 2 no softmax vector work at all (P = S), 3 no P.V MFMAs,
// 4 no S MFMAs, 5 no LDS-DMA staging behind the first two tiles, 6 no per-tile barrier, 7 no V^T LDS reads, 8 no K fragment reads;
// 9 is not an ablation: __launch_bounds__(256, 3) (results stay right)
#ifndef FWD_ABL
#define FWD_ABL 0
#endif
#ifdef FWD_STAMPS  // developer build (tools/dev/fwd_stamps.py, hipcc -DFWD_STAMPS): clock-tick sums per phase of a key tile, all active waves
__device__ unsigned long long fwd_dbg[16];
extern "C" void wft_fwd_dbg_read(unsigned long long* host, int reset) {
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fwd_dbg), z, sizeof z); return; }
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(fwd_dbg), 16 * sizeof(unsigned long long));
}
// stamp phase I behind the value DEP (a dependent v_mov makes the hardware wait for DEP's producer, e.g. an MFMA chain)
#define FWD_STAMP(I, DEP)                                                                                  \
  do {                                                                                                     \
    unsigned long long now_;                                                                               \
    asm volatile("v_mov_b32 %1, %1\n s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_), "+v"(DEP)); \
    fst[I] += now_ - flast;                                                                                \
    flast = now_;                                                                                          \
  } while (0)
#else
#define FWD_STAMP(I, DEP) do {} while (0)
#endif

// ------------------------------------------------------------------------------ forward
// K/V tiles travel through a THREE-slot LDS ring, staged two tiles ahead of their use, and the end-of-tile
// wait is a counted s_waitcnt vmcnt(4) (this wave's 4 LDS-DMA instructions of tile kt+2 may stay in flight).
// The transposed V reads are inline asm: with the builtin, hipcc drains every outstanding LDS-DMA
// (s_waitcnt vmcnt(0)) in front of the first ds_read_b64_tr of each tile, which cut the prefetch distance to
// half a tile and left the kernel latency-bound (no-load experiment: +27 %).
#if FWD_ABL == 9  // (occupancy experiment: three workgroups per CU = three waves per SIMD, <= 168 registers)
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnP p) {
#else
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p) {
#endif
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
  // (q_prescaled: c = 1.0 at run time.  A template instantiation without the multiplies measured 1 % SLOWER — 1 417 -> 1 432 us per
  // encoder call at 87 clips, three runs; the compiler's schedule, not the instruction count, decides here: profiles/r06_attn_prescale.md)
  const float c = p.c;
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
#ifdef FWD_STAMPS
  unsigned long long fst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flast = __builtin_amdgcn_s_memtime(), fbegin = flast;
#endif

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
    const bool more = FWD_ABL == 5 ? false : kt + 2 < nkt;
    if (more)
      att_stage2(stK, kb, p.ldk, smem + NXT2 * 16384, stV, vb, p.ldv, smem + NXT2 * 16384 + 8192, key0 + 128, p.Tk, wave, lane);
    // (a wave whose 32 queries all lie past the sequence end — T = 1500: the fourth wave of the last 128-query block — only
    // stages and keeps the barriers)
    const bool active = qw0 < p.Tq && !(p.causal && key0 > qw0 + 31);
    s16x4 vt[4][2][2];
    bf16x8 pf[4];
    FWD_STAMP(0, m);  // tile entry -> stage issue done
    if (active) {
      static_for<4>([&](auto ks_tag) {  // the k-step's 2048-byte stride rides in the immediate: no address add per read
        constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if constexpr (FWD_ABL == 7) vt[ks][db][t] = s16x4{(short)lane, 1, 2, 3};
            else vt[ks][db][t] = att_tr_asm<CUR * 16384 + 8192 + ks * 2048>(tra[db][t]);
          }
      });
      f32x16 sacc[2];
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        if constexpr (FWD_ABL == 4) {
          sacc[kb2] = minit;
#pragma unroll
          for (int s = 0; s < 4; ++s) sacc[kb2][s] += __builtin_bit_cast(float, (int)kf[kb2][s][0]) * 1e-30f;  // (keeps the K reads alive)
          continue;
        }
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
      FWD_STAMP(1, sacc[1][15]);  // V^T read issue + S MFMA chains complete
      float tmax = FWD_ABL == 2 ? 0.f : att_xhalf_max(att_max32(sacc[0], sacc[1]));  // max over the tile of S - m
      FWD_STAMP(2, tmax);  // maximum (in-lane tree + half exchange)
      if (FWD_ABL != 2 && (kt == 0 || __builtin_amdgcn_ballot_w64(tmax * c > ATT_TAU) != 0)) {
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
          if constexpr (FWD_ABL == 2) continue;
          const float p0 = FWD_ABL == 1 ? sacc[kb2][e] * c : __builtin_amdgcn_exp2f(sacc[kb2][e] * c);
          const float p1 = FWD_ABL == 1 ? sacc[kb2][e + 1] * c : __builtin_amdgcn_exp2f(sacc[kb2][e + 1] * c);
          ls0 += p0;
          ls1 += p1;
          sacc[kb2][e] = p0;
          sacc[kb2][e + 1] = p1;
        }
      l += att_xhalf_sum(ls0 + ls1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pf[ks] = att_pack8(sacc[ks >> 1], ks & 1);
      FWD_STAMP(3, l);  // (rescale branch,) exponentials, row sums, packs
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    FWD_STAMP(4, m);  // own LDS-DMA pieces of tile kt+1 landed, V^T fragments landed
    if constexpr (FWD_ABL != 6) __builtin_amdgcn_s_barrier();
    FWD_STAMP(5, m);  // barrier
    __builtin_amdgcn_sched_barrier(0);
    if (FWD_ABL != 8 && kt + 1 < nkt) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem + NXT * 16384, offs, kb2, s);
    }
    if (active) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          if constexpr (FWD_ABL == 3) {  // (keeps the V^T reads and the packs alive)
            oacc[db][ks] += __builtin_bit_cast(float, (int)vt[ks][db][0][0] | ((int)vt[ks][db][1][1] << 16)) * 1e-30f + __builtin_bit_cast(float, (int)pf[ks][0]) * 1e-30f;
            continue;
          }
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_join(vt[ks][db][0], vt[ks][db][1]), pf[ks], oacc[db], 0, 0, 0);
        }
      FWD_STAMP(6, oacc[1][15]);  // K fragment reads of tile kt+1 issued + P.V MFMA chains complete
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
    if (h == 0 && p.lse) p.lse[((long)b * p.H + hd) * p.Tq + qi] = m * p.ls + __logf(l);
  }
#ifdef FWD_STAMPS
  if (lane == 0 && qw0 < p.Tq) {
    for (int i = 0; i < 7; ++i) atomicAdd(&fwd_dbg[i], fst[i]);
    atomicAdd(&fwd_dbg[8], 1ull);
    atomicAdd(&fwd_dbg[9], __builtin_amdgcn_s_memtime() - fbegin);
    atomicAdd(&fwd_dbg[10], (unsigned long long)nkt);
  }
#endif
}


// Round 6: the softmax's vector section of the pipelined forward kernel runs at raised wave priority (s_setprio 1).  Two waves of
// DIFFERENT workgroups share a SIMD at an arbitrary phase; the arbiter is oldest-first, so without it the wave that is in its exponentials
// keeps losing issue slots to its partner's MFMA issue and both drift into phase.  Measured, alternating, three runs (encoder call at 87
// clips, prescaled q): 1 508 / 1 517 / 1 500 us -> 1 477 / 1 468 / 1 504 (-1.7 %); priority on the MFMA clusters instead: +-0.
// No arithmetic changes: bit-identical outputs (tests/test_attn_fwd_pipe_gpu.py).
#define FWD_PRIO_MFMA_ON
#define FWD_PRIO_MFMA_OFF
#define FWD_PRIO_VALU_ON __builtin_amdgcn_s_setprio(1)
#define FWD_PRIO_VALU_OFF __builtin_amdgcn_s_setprio(0)
// ------------------------------------------------------------------------------ forward, software-pipelined (round 5)
// The ablation builds of attn_fwd_kernel (FWD_ABL, profiles/r05_attn_fwd.md) behave like a SUM of their parts: taking out the S
// MFMAs saves 29 % of the kernel, the softmax's vector work 24 %, the P.V MFMAs 11 %, the LDS-DMA staging 13 %, the V^T reads
// 13 % — a wave issues its S chains and then sits on their results, and with the oldest-wave-first arbiter the second wave of the
// SIMD does not fill that hole reliably.  Here the S chains of tile kt+1 are issued BEHIND the softmax of tile kt and IN FRONT of
// its P.V chains: they run on the matrix pipe while the wave goes through the end-of-tile wait, the barrier, the next tile's
// staging and V^T reads, and tile kt+1's softmax finds its scores finished.  That needs tile kt+1's K fragments one barrier
// earlier, so the K/V ring has FOUR slots and is staged three tiles ahead (64 KiB per workgroup, two workgroups per CU):
//   iteration kt: stage kt+3 | V^T(kt) reads | softmax(kt) -> P | S(kt+1) MFMAs | wait own pieces of kt+2, barrier |
//                 K(kt+2) fragment reads | P.V(kt) MFMAs
// Same arithmetic in the same order as attn_fwd_kernel (the stale maximum that enters S(kt+1) as its initial accumulator is the
// one softmax(kt) has just settled, exactly what the un-pipelined kernel uses at the head of tile kt+1): bit-identical results.
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 16384];  // [slot 4][K 8K | V 8K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  // (q_prescaled: c = 1.0 at run time.  A template instantiation without the multiplies measured 1 % SLOWER — 1 417 -> 1 432 us per
  // encoder call at 87 clips, three runs; the compiler's schedule, not the instruction count, decides here: profiles/r06_attn_prescale.md)
  const float c = p.c;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = att_load_reg_frag(qrow, s, h);
  const AttOffs offs = att_offsets(lane);
  const AttStage stK = att_stage_init(p.ldk, wave, lane), stV = att_stage_init(p.ldv, wave, lane);
  const unsigned lds0 = lds_addr_of(smem);
  unsigned tra[2][2];
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
  float m = 0.f, l = 0.f;
  f32x16 minit = zero16;
  auto is_active = [&](int kt) { return qw0 < p.Tq && !(p.causal && kt * 64 > qw0 + 31); };

  // prologue: tiles 0, 1, 2 on their way; tiles 0 and 1 certified by the first barrier
  att_stage2(stK, kb, p.ldk, smem, stV, vb, p.ldv, smem + 8192, 0, p.Tk, wave, lane);
  if (nkt > 1) att_stage2(stK, kb, p.ldk, smem + 16384, stV, vb, p.ldv, smem + 16384 + 8192, 64, p.Tk, wave, lane);
  if (nkt > 2) {
    att_stage2(stK, kb, p.ldk, smem + 2 * 16384, stV, vb, p.ldv, smem + 2 * 16384 + 8192, 128, p.Tk, wave, lane);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 kf[2][4];  // K row fragments of the NEXT tile to be multiplied
  f32x16 sacc[2];   // scores of the CURRENT tile (S chains issued one tile ahead)
#pragma unroll
  for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem, offs, kb2, s);
  sacc[0] = sacc[1] = zero16;
  if (is_active(0)) {
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][0], qf[0], minit, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < 4; ++s) sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][s], qf[s], sacc[kb2], 0, 0, 0);
    }
  }
  if (nkt > 1) {
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem + 16384, offs, kb2, s);
  }

  auto tile = [&](auto cur_tag, int kt) {
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int NXT2 = (CUR + 2) % 4, NXT3 = (CUR + 3) % 4;
    const int key0 = kt * 64;
    const bool more = kt + 3 < nkt;
    if (more)
      att_stage2(stK, kb, p.ldk, smem + NXT3 * 16384, stV, vb, p.ldv, smem + NXT3 * 16384 + 8192, key0 + 192, p.Tk, wave, lane);
    const bool active = is_active(kt);
    s16x4 vt[4][2][2];
    bf16x8 pf[4];
    if (active) {
      static_for<4>([&](auto ks_tag) {
        constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int t = 0; t < 2; ++t) vt[ks][db][t] = att_tr_asm<CUR * 16384 + 8192 + ks * 2048>(tra[db][t]);
      });
      if ((key0 + 64 > p.Tk) || (p.causal && key0 + 63 > qw0)) {
        const int lim = (p.causal ? (qi + 1 < p.Tk ? qi + 1 : p.Tk) : p.Tk) - key0 - 4 * h;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            sacc[kb2][e] = (32 * kb2 + (e & 3) + 8 * (e >> 2)) < lim ? sacc[kb2][e] : ATT_NEG;
      }
      const float tmax = att_xhalf_max(att_max32(sacc[0], sacc[1]));
      if (kt == 0 || __builtin_amdgcn_ballot_w64(tmax * c > ATT_TAU) != 0) {
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
      FWD_PRIO_VALU_ON;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const f32x2 sc = f32x2{sacc[kb2][e], sacc[kb2][e + 1]} * c;  // (one v_pk_mul_f32 per pair: hipcc leaves the scalar form unpacked)
          const float p0 = __builtin_amdgcn_exp2f(sc[0]);
          const float p1 = __builtin_amdgcn_exp2f(sc[1]);
          ls0 += p0;
          ls1 += p1;
          sacc[kb2][e] = p0;
          sacc[kb2][e + 1] = p1;
        }
      l += att_xhalf_sum(ls0 + ls1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) pf[ks] = att_pack8(sacc[ks >> 1], ks & 1);
      FWD_PRIO_VALU_OFF;
    }
    // the NEXT tile's scores: on the matrix pipe from here, consumed by the next iteration's softmax
    if (kt + 1 < nkt && is_active(kt + 1)) {
      FWD_PRIO_MFMA_ON;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][0], qf[0], minit, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) sacc[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2][s], qf[s], sacc[kb2], 0, 0, 0);
      }
      FWD_PRIO_MFMA_OFF;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nkt) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[kb2][s] = att_row_frag(smem + NXT2 * 16384, offs, kb2, s);
    }
    if (active) {
      FWD_PRIO_MFMA_ON;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_join(vt[ks][db][0], vt[ks][db][1]), pf[ks], oacc[db], 0, 0, 0);
      FWD_PRIO_MFMA_OFF;
    }
  };
  int kt = 0;
  for (; kt + 3 < nkt; kt += 4) {
    tile(IntC<0>{}, kt);
    tile(IntC<1>{}, kt + 1);
    tile(IntC<2>{}, kt + 2);
    tile(IntC<3>{}, kt + 3);
  }
  if (kt < nkt) tile(IntC<0>{}, kt);
  if (kt + 1 < nkt) tile(IntC<1>{}, kt + 1);
  if (kt + 2 < nkt) tile(IntC<2>{}, kt + 2);

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
    if (h == 0 && p.lse) p.lse[((long)b * p.H + hd) * p.Tq + qi] = m * p.ls + __logf(l);
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
  const float nlse = -p.lse[sidx] / p.ls;
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
  const float c = p.c;
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
  const float c = p.c;
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
        u32x2 pk = {pack2bf(dkacc[db][4 * a] * p.ls, dkacc[db][4 * a + 1] * p.ls),
                    pack2bf(dkacc[db][4 * a + 2] * p.ls, dkacc[db][4 * a + 3] * p.ls)};
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
// ------------------------------------------------------------------------------ dK/dV, one wave per SIMD (round 4)
// The non-causal dK/dV sweep (the encoder's 1500 x 1500 attention and the decoder's cross attention: 85 of a step's 657 ms in
// the kernel above, matrix pipe 29 % busy) as ONE hand-scheduled stream per SIMD, like gemm_nt4w.hip.  In the kernel above a
// wave's chain  S / dP MFMAs -> exp, multiplies, packing -> dV / dK MFMAs  is serial and only the SIMD's other wave fills the
// holes; here a wave owns 64 keys (two 32-key blocks, K / V fragments in a[128:191]) and runs THREE query blocks at once:
//   iteration j:  MFMA  slots  0-15  dV / dK of block j-1   (operands: P, dS packed in v[32:63]; Q^T, dO^T fragments v[96:127])
//                       slots 16-31  S / dP of block j+1    (into the other accumulator generation, which STARTS from the row
//                                                            constants -lse/scale and -delta read from LDS as the C operand)
//                 VALU  4 per slot   block j: slots 0-15  p = exp2(c S')  (2 multiplies + 2 exponentials per slot),
//                                             slots 16-23 dS = p dP',  slots 24-31 the 32 bf16 packs
//                 LDS   slots  0-11  row fragments of Q, dO and the constants of block j+1;  slots 16-31 transposed fragments of j
// so the matrix pipe never waits for the vector stream of the same block (budget per 32-cycle MFMA: 24 issue cycles; 2 exp + 2
// mul = 24).  Q / dO fragments and constants are shared by the wave's two key blocks (half the LDS traffic per MFMA).
// Queries beyond Tq need no masks: their Q / dO rows and constants arrive as ZEROS (buffer descriptors that end at row Tq), so
// they add exp2(0) * 0 = 0 to dV and 0 * finite = 0 to dK; keys beyond Tk are lanes whose results are not stored.
// LDS: three buffers of one 64-query tile {Q, dO: 8 pieces of 8 rows, 1 280 B apart, piece pid at pid * 1280 + 64 (pid & 1) +
// 16 (pid >> 1); -lse/scale, -delta: 256 B each}.  Piece pid holds queries q0 + {0, 2} + 16 m, q0 = (pid & 1) + 4 (pid >> 1): row q sits
// at 64 (q & 3) + 16 ((q >> 2) & 3) modulo 256 B, which makes BOTH the 32-row ds_read_b128 operand reads and the 4-row
// ds_read_b64_tr_b16 reads bank-conflict-free with every fragment address = one base register + an immediate.
#define D4_PIECE 1280
#define D4_DO 10240
#define D4_LSE 20480
#define D4_DLT 20736
#define D4_BUF 20992
#define D4_KV 20480                      // one wave's K and V fragments in transit: two tiles of 10 240 B
#define D4_LDS (3 * D4_BUF + 4 * D4_KV)
#define D4_STR2(x) #x
#define D4_STR(x) D4_STR2(x)

#ifndef D4_EXP
#define D4_EXP 0  // developer timing experiments (results wrong): 1 no LDS-DMA in the loop, 2 no vector arithmetic, 3 no LDS reads, 4 no barrier
#endif
#define D4_ASM_MACROS ".set d4_exp, " D4_STR(D4_EXP) "\n" R"ASM(
.macro D4_VALU ops:vararg
  .if d4_exp == 10
    v_mov_b32 v29, v30
  .elseif d4_exp == 11
    s_nop 0
  .elseif d4_exp == 12
    v_mov_b32 v29, s66
  .elseif d4_exp != 2 && !(d4_exp == 8 && d4_s < 16) && !(d4_exp == 9 && d4_s >= 16)
    \ops
  .endif
.endm
.macro D4_EXPF d, x
  .if d4_exp == 10
    v_mov_b32 v29, v30
  .elseif d4_exp == 11
    s_nop 0
  .elseif d4_exp == 12
    v_mov_b32 v29, s66
  .elseif d4_exp == 5
    v_mov_b32 \d, \x
  .elseif d4_exp == 6
  .elseif d4_exp != 2 && d4_exp != 8 && d4_exp < 13
    v_exp_f32 \d, \x
  .endif
.endm
; registers: S(g,kb) v[g+32kb..+15], dP(g,kb) v[g+16+32kb..+15], g = 128 / 192;  PF(kb) v[32+16kb..+7], DSF(kb) v[40+16kb..+7];
; row constants v[64:79] (-lse/scale), v[80:95] (-delta); Q rows AQ(s) a[192+4s..], dO rows AD(s) a[208+4s..]; TQ(ks,db) v[96+8ks+4db..], TD(ks,db) v[112+8ks+4db..];
; dK(kb,db) a[64kb+16db..+15], dV(kb,db) a[64kb+32+16db..+15]; KF(kb,s) a[128+32kb+4s..+3], VF(kb,s) a[144+32kb+4s..+3]
.macro D4_M2 n
  .if ((\n) %% 2) == 0
    v_mfma_f32_32x32x16_bf16 a[64*((\n)/8)+32+16*(((\n)/2)%%2):64*((\n)/8)+32+16*(((\n)/2)%%2)+15], v[112+8*(((\n)/4)%%2)+4*(((\n)/2)%%2):112+8*(((\n)/4)%%2)+4*(((\n)/2)%%2)+3], v[32+16*((\n)/8)+4*(((\n)/4)%%2):32+16*((\n)/8)+4*(((\n)/4)%%2)+3], a[64*((\n)/8)+32+16*(((\n)/2)%%2):64*((\n)/8)+32+16*(((\n)/2)%%2)+15]
  .else
    v_mfma_f32_32x32x16_bf16 a[64*((\n)/8)+16*(((\n)/2)%%2):64*((\n)/8)+16*(((\n)/2)%%2)+15], v[96+8*(((\n)/4)%%2)+4*(((\n)/2)%%2):96+8*(((\n)/4)%%2)+4*(((\n)/2)%%2)+3], v[40+16*((\n)/8)+4*(((\n)/4)%%2):40+16*((\n)/8)+4*(((\n)/4)%%2)+3], a[64*((\n)/8)+16*(((\n)/2)%%2):64*((\n)/8)+16*(((\n)/2)%%2)+15]
  .endif
.endm
.macro D4_M1 n, g
  ; chain (\n)/4: 0 S kb0, 1 dP kb0, 2 S kb1, 3 dP kb1; k-step (\n)%%4.  The first MFMA of a chain starts from the block's row
  ; constants (v[64:79] -lse/scale for S, v[80:95] -delta for dP), shared by the two key blocks
  .if ((\n) %% 4) == 0
    v_mfma_f32_32x32x16_bf16 v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15], a[192+16*(((\n)/4)%%2):192+16*(((\n)/4)%%2)+3], a[128+16*(((\n)/4)%%2)+32*((\n)/8):128+16*(((\n)/4)%%2)+32*((\n)/8)+3], v[64+16*(((\n)/4)%%2):64+16*(((\n)/4)%%2)+15]
  .else
    v_mfma_f32_32x32x16_bf16 v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15], a[192+16*(((\n)/4)%%2)+4*((\n)%%4):192+16*(((\n)/4)%%2)+4*((\n)%%4)+3], a[128+16*(((\n)/4)%%2)+32*((\n)/8)+4*((\n)%%4):128+16*(((\n)/4)%%2)+32*((\n)/8)+4*((\n)%%4)+3], v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15]
  .endif
.endm
; row read i (0..15) of a block, in the order the S / dP chains need them: 0-3 -lse/scale -> v[64:79], 4-7 Q rows -> a[192:207],
; 8-11 -delta -> v[80:95], 12-15 dO rows -> a[208:223]
.macro D4_RD1 i, rb, cb, qb
  .if d4_exp == 3
  .elseif (\i) < 4
    ds_read_b128 v[64+4*(\i):64+4*(\i)+3], \cb offset:20480+128*\qb+32*(\i)
  .elseif (\i) < 8
    ds_read_b128 a[192+4*((\i)-4):192+4*((\i)-4)+3], \rb offset:512*\qb+32*((\i)-4)
  .elseif (\i) < 12
    ds_read_b128 v[80+4*((\i)-8):80+4*((\i)-8)+3], \cb offset:20736+128*\qb+32*((\i)-8)
  .else
    ds_read_b128 a[208+4*((\i)-12):208+4*((\i)-12)+3], \rb offset:10240+512*\qb+32*((\i)-12)
  .endif
.endm
; transposed read m (0..15) in the order the dV / dK MFMAs consume them: (ks, db) = (m/8, (m/4)%2); (m/2)%2 = 0 dO^T, 1 Q^T; t = m%2
.macro D4_RD2 m, tb, qb
  .if d4_exp != 3
  ds_read_b64_tr_b16 v[112-16*(((\m)/2)%%2)+8*((\m)/8)+4*(((\m)/4)%%2)+2*((\m)%%2):112-16*(((\m)/2)%%2)+8*((\m)/8)+4*(((\m)/4)%%2)+2*((\m)%%2)+1], \tb offset:10240*(1-((\m)/2)%%2)+5152*((\m)%%2)+128*(4*\qb+2*((\m)/8))+64*(((\m)/4)%%2)
  .endif
.endm
; next 64-query tile: source bases += 64 rows (Q, dO) / 256 B (constants), bounds shrink with them (not below zero)
.macro D4_ADVANCE
  s_add_u32 s40, s40, s56
  s_addc_u32 s41, s41, 0
  s_sub_u32 s42, s42, s56
  s_cselect_b32 s42, 0, s42
  s_add_u32 s44, s44, s57
  s_addc_u32 s45, s45, 0
  s_sub_u32 s46, s46, s57
  s_cselect_b32 s46, 0, s46
  s_add_u32 s48, s48, 256
  s_addc_u32 s49, s49, 0
  s_sub_u32 s50, s50, 256
  s_cselect_b32 s50, 0, s50
.endm
; step i (0..4) of this wave's share of one tile -> the buffer at LDS offset \boff (an SGPR): pieces 2 wave, 2 wave + 1 of Q (0, 1)
; and dO (2, 3); 4: the constants (even waves -lse/scale, odd waves -delta; waves 2, 3 repeat 0, 1 so that every wave has
; five pieces per tile in flight and the counted waits are uniform)
.macro D4_DMA i, boff
  .if (\i) == 0
    s_add_u32 m0, s58, \boff
    s_nop 0
    buffer_load_dwordx4 %[voQ0], s[40:43], 0 offen lds
  .elseif (\i) == 1
    s_add_u32 m0, s59, \boff
    s_nop 0
    buffer_load_dwordx4 %[voQ1], s[40:43], 0 offen lds
  .elseif (\i) == 2
    s_add_u32 m0, s58, \boff
    s_add_u32 m0, m0, 10240
    s_nop 0
    buffer_load_dwordx4 %[voD0], s[44:47], 0 offen lds
  .elseif (\i) == 3
    s_add_u32 m0, s59, \boff
    s_add_u32 m0, m0, 10240
    s_nop 0
    buffer_load_dwordx4 %[voD1], s[44:47], 0 offen lds
  .else
    s_add_u32 m0, s60, \boff
    s_nop 0
    buffer_load_dword %[voC], s[48:51], 0 offen lds
  .endif
.endm
.macro D4_STAGE boff
  D4_DMA 0, \boff
  D4_DMA 1, \boff
  D4_DMA 2, \boff
  D4_DMA 3, \boff
  D4_DMA 4, \boff
.endm
; one iteration.  gV: generation (128 / 192) whose block is exponentiated here, gM: the other (target of the S / dP MFMAs);
; rb1, cb1, qb1: bases / half of the block whose row fragments are read; tb2, qb2: of the block whose transposed fragments are read;
; dma = 1: this wave's LDS-DMA share of the tile two ahead goes out in slots 16-20 (buffer offset s65), sources advance in slot 21.
; A wave alone on its SIMD issues in order and an MFMA gap hides about 24 issue cycles (measured here: two junk v_mov per gap cost
; 1.4 cycles each, the third and fourth 3.4): every slot carries 20 cycles of vector work and ONE LDS read —
;   slots  0-15  c-multiply of key block 1, both exponentials; row read number slot
;   slots 16-23  four dS multiplies, one pack; transposed read         24-31  three packs, the c-multiplies of the NEXT block's
;                key block 0 (its S chain finished in slot 19); transposed read
; LDS reads are counted, never drained: transposed fragment f (two reads, slots 16 + 2 f, 17 + 2 f) is consumed by slot f of the
; next iteration behind lgkmcnt(14 - f) (the later transposed reads + the f row reads issued since); the row reads behind
; lgkmcnt(8) (slot 16: constants and Q rows) and lgkmcnt(4) (slot 20: -delta, dO rows; four transposed reads are younger).
.macro D4_ITER gV, gM, rb1, cb1, qb1, tb2, qb2, dma
  .set d4_s, 0
  .rept 32
    .if d4_s < 8 && d4_exp != 7
      s_waitcnt lgkmcnt(14-d4_s)
    .elseif d4_s == 16
      s_waitcnt lgkmcnt(8)
    .elseif d4_s == 20
      s_waitcnt lgkmcnt(4)
    .endif
    .if d4_s < 16
      D4_M2 d4_s
      .if att_pre == 0
      D4_VALU v_mul_f32 v[\gV+32+d4_s], %[c], v[\gV+32+d4_s]
      .endif
      D4_EXPF v[\gV+d4_s], v[\gV+d4_s]
      D4_RD1 d4_s, \rb1, \cb1, \qb1
      D4_EXPF v[\gV+32+d4_s], v[\gV+32+d4_s]
    .else
      D4_M1 d4_s-16, \gM
      .if d4_s < 24
        D4_VALU v_mul_f32 v[\gV+16+2*(d4_s-16)], v[\gV+2*(d4_s-16)], v[\gV+16+2*(d4_s-16)]
        D4_VALU v_mul_f32 v[\gV+16+2*(d4_s-16)+1], v[\gV+2*(d4_s-16)+1], v[\gV+16+2*(d4_s-16)+1]
        D4_RD2 d4_s-16, \tb2, \qb2
        D4_VALU v_mul_f32 v[\gV+48+2*(d4_s-16)], v[\gV+32+2*(d4_s-16)], v[\gV+48+2*(d4_s-16)]
        D4_VALU v_mul_f32 v[\gV+48+2*(d4_s-16)+1], v[\gV+32+2*(d4_s-16)+1], v[\gV+48+2*(d4_s-16)+1]
        D4_VALU v_cvt_pk_bf16_f32 v[32+(d4_s-16)], v[\gV+2*(d4_s-16)], v[\gV+2*(d4_s-16)+1]
        .if \dma && d4_s < 21 && d4_exp != 1
          D4_DMA d4_s-16, s65
        .endif
        .if \dma && d4_s == 21
          D4_ADVANCE
        .endif
      .else
        D4_VALU v_cvt_pk_bf16_f32 v[48+(d4_s-24)], v[\gV+32+2*(d4_s-24)], v[\gV+32+2*(d4_s-24)+1]
        D4_VALU v_cvt_pk_bf16_f32 v[40+(d4_s-24)], v[\gV+16+2*(d4_s-24)], v[\gV+16+2*(d4_s-24)+1]
        D4_RD2 d4_s-16, \tb2, \qb2
        D4_VALU v_cvt_pk_bf16_f32 v[56+(d4_s-24)], v[\gV+48+2*(d4_s-24)], v[\gV+48+2*(d4_s-24)+1]
        .if att_pre == 0
        D4_VALU v_mul_f32 v[\gM+2*(d4_s-24)], %[c], v[\gM+2*(d4_s-24)]
        D4_VALU v_mul_f32 v[\gM+2*(d4_s-24)+1], %[c], v[\gM+2*(d4_s-24)+1]
        .endif
      .endif
    .endif
    .set d4_s, d4_s+1
  .endr
.endm
)ASM"

#define D4_ASM_PURGE R"ASM(
.purgem D4_VALU
.purgem D4_EXPF
.purgem D4_M2
.purgem D4_M1
.purgem D4_RD1
.purgem D4_RD2
.purgem D4_ADVANCE
.purgem D4_DMA
.purgem D4_STAGE
.purgem D4_ITER
)ASM"

#define D4_A8(x) "a" D4_STR(x##0), "a" D4_STR(x##1), "a" D4_STR(x##2), "a" D4_STR(x##3), "a" D4_STR(x##4), "a" D4_STR(x##5), "a" D4_STR(x##6), "a" D4_STR(x##7), "a" D4_STR(x##8), "a" D4_STR(x##9)
#define D4_V8(x) "v" D4_STR(x##0), "v" D4_STR(x##1), "v" D4_STR(x##2), "v" D4_STR(x##3), "v" D4_STR(x##4), "v" D4_STR(x##5), "v" D4_STR(x##6), "v" D4_STR(x##7), "v" D4_STR(x##8), "v" D4_STR(x##9)
#define D4_S8(x) "s" D4_STR(x##0), "s" D4_STR(x##1), "s" D4_STR(x##2), "s" D4_STR(x##3), "s" D4_STR(x##4), "s" D4_STR(x##5), "s" D4_STR(x##6), "s" D4_STR(x##7), "s" D4_STR(x##8), "s" D4_STR(x##9)
// a0..a223, v24..v255, s40..s79
#define D4_CLOBBER_A "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", D4_A8(1), D4_A8(2), D4_A8(3), D4_A8(4), D4_A8(5), D4_A8(6), D4_A8(7), D4_A8(8), D4_A8(9), D4_A8(10), D4_A8(11), D4_A8(12), D4_A8(13), D4_A8(14), D4_A8(15), D4_A8(16), D4_A8(17), D4_A8(18), D4_A8(19), D4_A8(20), D4_A8(21), "a220", "a221", "a222", "a223"
#define D4_CLOBBER_V "v24", "v25", "v26", "v27", "v28", "v29", D4_V8(3), D4_V8(4), D4_V8(5), D4_V8(6), D4_V8(7), D4_V8(8), D4_V8(9), D4_V8(10), D4_V8(11), D4_V8(12), D4_V8(13), D4_V8(14), D4_V8(15), D4_V8(16), D4_V8(17), D4_V8(18), D4_V8(19), D4_V8(20), D4_V8(21), D4_V8(22), D4_V8(23), D4_V8(24), "v250", "v251", "v252", "v253", "v254", "v255"
#define D4_CLOBBER_S D4_S8(4), D4_S8(5), D4_S8(6), D4_S8(7)

template <int N>
__device__ __forceinline__ float d4_acc() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "i"(N));
  return x;
}
template <int BASE>
__device__ __forceinline__ f32x16 d4_get16() {
  f32x16 v;
  v[0] = d4_acc<BASE>(); v[1] = d4_acc<BASE + 1>(); v[2] = d4_acc<BASE + 2>(); v[3] = d4_acc<BASE + 3>();
  v[4] = d4_acc<BASE + 4>(); v[5] = d4_acc<BASE + 5>(); v[6] = d4_acc<BASE + 6>(); v[7] = d4_acc<BASE + 7>();
  v[8] = d4_acc<BASE + 8>(); v[9] = d4_acc<BASE + 9>(); v[10] = d4_acc<BASE + 10>(); v[11] = d4_acc<BASE + 11>();
  v[12] = d4_acc<BASE + 12>(); v[13] = d4_acc<BASE + 13>(); v[14] = d4_acc<BASE + 14>(); v[15] = d4_acc<BASE + 15>();
  return v;
}

#ifdef D4_STAMPS
#define D4_STAMP_ASM(r) "s_memtime s[" #r ":" #r "+1]\n s_waitcnt lgkmcnt(0)\n"
#else
#define D4_STAMP_ASM(r) ""
#endif
#ifdef D4_STAMPS  // developer build (tools/dev/dkdv4w_stamps.py): clock-tick sums over workgroups, wave 0
__device__ unsigned long long d4_dbg[16];
extern "C" void wft_dbg_read(unsigned long long* host, int reset) {
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(d4_dbg), z, sizeof z); return; }
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(d4_dbg), 16 * sizeof(unsigned long long));
}
#endif
// descriptors and requests shared by the main block and the prefetch block of attn_bwd_dkdv4w_kernel
#define D4_ASM_MACROS2 R"ASM(
; descriptors: Q s[40:43], dO s[44:47], this wave's constants (-lse/scale or -delta) s[48:51], K s[68:71], V s[72:75];
; tile strides s56, s57; LDS-DMA destinations inside a buffer: piece 2 wave (s58), 2 wave + 1 (s59), this wave's constant row (s60:
; even waves lse, odd delta); buffer offsets: cur (tile T) s63, nxt (T+1) s64, ld (T+2) s65; this wave's K / V transit area s52
.macro D4_SRD_INIT
  s_mov_b64 s[40:41], %[bQ]
  s_lshr_b32 s61, %[stQ], 6
  s_sub_u32 s62, %[tq], 1
  s_mul_i32 s42, s62, s61
  s_add_u32 s42, s42, 128
  s_mov_b32 s43, 0x20000
  s_mov_b64 s[44:45], %[bD]
  s_lshr_b32 s61, %[stD], 6
  s_mul_i32 s46, s62, s61
  s_add_u32 s46, s46, 128
  s_mov_b32 s47, 0x20000
  s_and_b32 s61, %[wave], 1
  s_cmp_eq_u32 s61, 0
  s_cselect_b64 s[48:49], %[bL], %[bT]
  s_lshl_b32 s50, %[tq], 2
  s_mov_b32 s51, 0x20000
  s_mov_b32 s56, %[stQ]
  s_mov_b32 s57, %[stD]
  s_mov_b64 s[68:69], %[bK]
  s_sub_u32 s62, %[tk], 1
  s_mul_i32 s70, s62, %[ldk2]
  s_add_u32 s70, s70, 128
  s_mov_b32 s71, 0x20000
  s_mov_b64 s[72:73], %[bV]
  s_mul_i32 s74, s62, %[ldv2]
  s_add_u32 s74, s74, 128
  s_mov_b32 s75, 0x20000
  s_lshl_b32 s61, %[wave], 1
  s_mul_i32 s58, s61, 1280
  s_lshr_b32 s62, s61, 1
  s_lshl_b32 s62, s62, 4
  s_add_u32 s58, s58, s62
  s_add_u32 s58, s58, %[lds0]
  s_add_u32 s59, s58, 1344
  s_and_b32 s60, %[wave], 1
  s_mul_i32 s60, s60, 256
  s_add_u32 s60, s60, 20480
  s_add_u32 s60, s60, %[lds0]
  s_mov_b32 s63, 0
  s_mov_b32 s64, )ASM" D4_STR(D4_BUF) R"ASM(
  s_mov_b32 s65, 2*)ASM" D4_STR(D4_BUF) R"ASM(
  s_mul_i32 s52, %[wave], )ASM" D4_STR(D4_KV) R"ASM(
  s_add_u32 s52, s52, 3*)ASM" D4_STR(D4_BUF) R"ASM(
  s_add_u32 s52, s52, %[lds0]
.endm
; K / V rows of this wave's 64 keys -> its transit area, as two tiles in the piece layout of the Q / dO tiles (8 + 8 LDS-DMA pieces of
; eight whole 128-byte rows: the row-per-lane fragment loads they replace touched every line four times and cost ~400 cycles of
; issue each).  Piece pid holds keys q0 + {0, 2} + 16 m, q0 = (pid & 1) + 4 (pid >> 1); rows past Tk lie beyond the descriptors: zeros.
.macro D4_KVDMA
  .set d4_i, 0
  .rept 8
    s_mul_i32 s61, %[ldk2], (d4_i%%2)+4*(d4_i/2)
    v_add_u32 v29, s61, %[voKp]
    s_add_u32 m0, s52, d4_i*1280+64*(d4_i%%2)+16*(d4_i/2)
    s_mul_i32 s62, %[ldv2], (d4_i%%2)+4*(d4_i/2)
    buffer_load_dwordx4 v29, s[68:71], 0 offen lds
    v_add_u32 v30, s62, %[voVp]
    s_add_u32 m0, s52, 10240+d4_i*1280+64*(d4_i%%2)+16*(d4_i/2)
    s_nop 0
    buffer_load_dwordx4 v30, s[72:75], 0 offen lds
    .set d4_i, d4_i+1
  .endr
.endm
; ... and from there into a[128:191] as row fragments (lane (r, h): key 32 kb + r, columns 16 s + 8 h .. + 8)
.macro D4_KVRD
  s_sub_u32 s61, s52, %[lds0]
  v_add_u32 v29, s61, %[rb]
  .set d4_i, 0
  .rept 8
    ds_read_b128 a[128+32*(d4_i/4)+4*(d4_i%%4):128+32*(d4_i/4)+4*(d4_i%%4)+3], v29 offset:512*(d4_i/4)+32*(d4_i%%4)
    ds_read_b128 a[144+32*(d4_i/4)+4*(d4_i%%4):144+32*(d4_i/4)+4*(d4_i%%4)+3], v29 offset:10240+512*(d4_i/4)+32*(d4_i%%4)
    .set d4_i, d4_i+1
  .endr
.endm
; first two tiles of an item -> buffers 0, 1 (sources end up two tiles on)
.macro D4_STAGE2
  D4_STAGE s63
  D4_ADVANCE
  s_nop 4
  D4_STAGE s64
  D4_ADVANCE
.endm
)ASM"
#define D4_ASM_PURGE2 R"ASM(
.purgem D4_SRD_INIT
.purgem D4_KVDMA
.purgem D4_KVRD
.purgem D4_STAGE2
)ASM"

template <bool PRE>  // PRE: q_prescaled — the asm loops are assembled without their c-scale multiplies (.set att_pre)
__global__ __launch_bounds__(256) void attn_bwd_dkdv4w_kernel(AttnP p) {
#ifdef D4_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // PERSISTENT: one workgroup per CU (it owns the CU: one wave per SIMD, 512 registers) walks work items (batch, head, 256-key
  // block).  The workgroups of one XCD take that XCD's items round-robin, in the order (head group, key block): the ~32 items
  // in flight on an XCD are the key blocks of 5-6 heads, whose Q / dO tiles (384 KB per head) stay in that XCD's L2, exactly
  // as with one workgroup per item (att_block_coords).  The next item's K / V fragments and first three tiles are requested
  // before this item's epilogue, so only a workgroup's first item waits for memory.
  const int nkb = (p.Tk + 255) >> 8;
  const int ngrp = p.H * p.B;
  const bool xcd_mode = ((ngrp & 7) == 0) && p.xcd && ((gridDim.x & 7) == 0);
  const int xcd = xcd_mode ? (int)(blockIdx.x & 7) : 0;
  const int w0 = xcd_mode ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wstep = xcd_mode ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int nitems = (xcd_mode ? ngrp >> 3 : ngrp) * nkb;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // fragment read bases (see the layout note above)
  const int c = r & 15, pidr = (c & 1) | ((c >> 2) << 1);
  const unsigned rb = lds0 + pidr * D4_PIECE + 64 * (pidr & 1) + 16 * (pidr >> 1) + (2 * (r >> 4) + ((c >> 1) & 1)) * 128 + h * 16;
  const int g4 = lane >> 4, i16 = lane & 15, pidt = ((i16 >> 2) & 1) | ((g4 >> 1) << 1);
  const unsigned tb = lds0 + pidt * D4_PIECE + 64 * (pidt & 1) + 16 * (pidt >> 1) + ((i16 >> 3) & 1) * 128 + 32 * (g4 & 1) + 8 * (i16 & 3);
  const unsigned cb = lds0 + 16 * h;
  // LDS-DMA share of this wave: pieces 2 wave and 2 wave + 1 of Q and of dO; lane: slot lane >> 3 (query q0 + 2 (slot & 1) + 16 (slot >> 1)), chunk lane & 7
  const int slot = lane >> 3, ch = lane & 7;
  auto qrow = [&](int pid) { return (pid & 1) + 4 * (pid >> 1) + 2 * (slot & 1) + 16 * (slot >> 1); };
  const unsigned voQ0 = (unsigned)(qrow(2 * wave) * (int)p.ldq + ch * 8) * 2u, voQ1 = (unsigned)(qrow(2 * wave + 1) * (int)p.ldq + ch * 8) * 2u;
  const unsigned voD0 = (unsigned)(qrow(2 * wave) * (int)p.lddo + ch * 8) * 2u, voD1 = (unsigned)(qrow(2 * wave + 1) * (int)p.lddo + ch * 8) * 2u;
  const unsigned voC = (unsigned)lane * 4u;
  auto sg64 = [](unsigned long long x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const unsigned tq = __builtin_amdgcn_readfirstlane((unsigned)p.Tq), tk = __builtin_amdgcn_readfirstlane((unsigned)p.Tk);
  const unsigned ldk2 = __builtin_amdgcn_readfirstlane((unsigned)p.ldk * 2u), ldv2 = __builtin_amdgcn_readfirstlane((unsigned)p.ldv * 2u);
  const unsigned stQ = __builtin_amdgcn_readfirstlane((unsigned)p.ldq * 128u), stD = __builtin_amdgcn_readfirstlane((unsigned)p.lddo * 128u);
  const unsigned npair = __builtin_amdgcn_readfirstlane((unsigned)((p.Tq + 63) >> 6));  // 64-query tiles = iteration pairs
  const float cscale = p.c;
  const unsigned cbits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, cscale));
  const unsigned wv = (unsigned)wave;
  // everything that depends on the work item
  struct Item {
    int b, hd, kw0;
    unsigned long long bQ, bD, bL, bT, bK, bV;
    unsigned voKp, voVp;
  };
  auto item = [&](int t) {
    Item x;
    const int g = xcd_mode ? (t / nkb) * 8 + xcd : t / nkb;
    x.hd = g % p.H;
    x.b = g / p.H;
    x.kw0 = (t % nkb) * 256 + wave * 64;
    const long sbase = ((long)x.b * p.H + x.hd) * p.Tq;
    x.bQ = sg64((unsigned long long)(p.q + (long)x.b * p.q_bs + x.hd * 64));
    x.bD = sg64((unsigned long long)(p.d_o + (long)x.b * p.do_bs + x.hd * 64));
    x.bL = sg64((unsigned long long)(p.delta + (long)p.B * p.H * p.Tq + sbase));  // -lse / scale (written by the dQ kernel)
    x.bT = sg64((unsigned long long)(p.delta + sbase));                            // -delta
    x.bK = sg64((unsigned long long)(p.k + (long)x.b * p.k_bs + x.hd * 64));
    x.bV = sg64((unsigned long long)(p.v + (long)x.b * p.v_bs + x.hd * 64));
    // byte offset of this lane's share of K / V piece 0 of the wave's 64 keys (slot lane >> 3: key 2 (slot & 1) + 16 (slot >> 1),
    // chunk lane & 7), relative to the (batch, head) bases
    x.voKp = (unsigned)((x.kw0 + 2 * (slot & 1) + 16 * (slot >> 1)) * (int)p.ldk + ch * 8) * 2u;
    x.voVp = (unsigned)((x.kw0 + 2 * (slot & 1) + 16 * (slot >> 1)) * (int)p.ldv + ch * 8) * 2u;
    return x;
  };

  for (int t = w0; t < nitems; t += wstep) {
  const Item cur = item(t);
  const int b = cur.b, hd = cur.hd, kw0 = cur.kw0;
  const unsigned first = __builtin_amdgcn_readfirstlane((unsigned)(t == w0));

  asm volatile(".set att_pre, %c[pre]\n" D4_ASM_MACROS D4_ASM_MACROS2 R"ASM(
    D4_SRD_INIT
    s_mov_b32 s66, %[npair]      ; loop counter
    s_cmp_eq_u32 %[first], 0
    s_cbranch_scc1 2f
    ; ---- first item of this workgroup: its K / V rows and tiles 0, 1 are requested here ...
    D4_KVDMA
    D4_STAGE2
    s_branch 3f
2:
    ; ---- ... later ones found them requested by the prefetch block behind the previous item (below): only the descriptors move on
    D4_ADVANCE
    D4_ADVANCE
    s_waitcnt vmcnt(0)
3:
    ; ---- (under the loads) accumulators, packed operands and transposed fragments start from zero: iteration 0 multiplies them
    .set d4_i, 0
    .rept 128
      v_accvgpr_write_b32 a[d4_i], 0
      .set d4_i, d4_i+1
    .endr
    .set d4_i, 32
    .rept 32
      v_mov_b32 v[d4_i], 0
      v_mov_b32 v[d4_i+64], 0
      .set d4_i, d4_i+1
    .endr
    v_mov_b32 v24, %[rb]
    v_mov_b32 v25, %[tb]
    v_mov_b32 v26, %[cb]
    v_add_u32 v27, s64, v24
    v_add_u32 v28, s64, v26
    s_waitcnt vmcnt(5)         ; K / V rows and tile 0 (tile 1: five pieces may still be in flight)
    s_barrier
    D4_KVRD
    )ASM" D4_STAMP_ASM(76) R"ASM(
    ; ---- block 0: row fragments + constants, S / dP -> generation 128
    .set d4_i, 0
    .rept 16
      D4_RD1 d4_i, v24, v26, 0
      .set d4_i, d4_i+1
    .endr
    s_waitcnt lgkmcnt(0)
    .set d4_i, 0
    .rept 16
      D4_M1 d4_i, 128
      .set d4_i, d4_i+1
    .endr
    s_nop 15
    s_nop 15
    .set d4_i, 0
    .if att_pre == 0
    .rept 16
      v_mul_f32 v[128+d4_i], %[c], v[128+d4_i]
      .set d4_i, d4_i+1
    .endr
    .endif
1:
    ; ==== tile boundary: tile T+1 has landed for every wave, tile T-1's buffer is free -> tile T+2 goes into it during this
    ; iteration (a fourth buffer and three tiles of distance measured the same; the space carries the K / V transit areas)
    .if d4_exp != 4
    s_waitcnt vmcnt(0)
    s_barrier
    .endif
    ; even iteration (block 2T): row fragments of block 2T+1 (this tile, half 1), transposed fragments of block 2T (half 0)
    D4_ITER 128, 192, v24, v26, 1, v25, 0, 1
    ; odd iteration (block 2T+1): row fragments of block 2T+2 (next tile, half 0), transposed fragments of block 2T+1
    D4_ITER 192, 128, v27, v28, 0, v25, 1, 0
    ; rotate the buffers: cur <- nxt <- ld <- cur
    s_mov_b32 s67, s63
    s_mov_b32 s63, s64
    s_mov_b32 s64, s65
    s_mov_b32 s65, s67
    v_add_u32 v24, s63, %[rb]
    v_add_u32 v25, s63, %[tb]
    v_add_u32 v26, s63, %[cb]
    v_add_u32 v27, s64, %[rb]
    v_add_u32 v28, s64, %[cb]
    s_sub_u32 s66, s66, 1
    s_cmp_eq_u32 s66, 0
    s_cbranch_scc0 1b
    ; ---- dV / dK of the last block
    s_waitcnt lgkmcnt(0)
    )ASM" D4_STAMP_ASM(78) R"ASM(
    s_nop 1
    .set d4_i, 0
    .rept 16
      D4_M2 d4_i
      .set d4_i, d4_i+1
    .endr
    s_waitcnt vmcnt(0)         ; (the tiles requested past the last query block: zeros, but they must not land later)
    s_nop 15
  )ASM" D4_ASM_PURGE D4_ASM_PURGE2
               :
               : [rb] "v"(rb), [tb] "v"(tb), [cb] "v"(cb), [voQ0] "v"(voQ0), [voQ1] "v"(voQ1), [voD0] "v"(voD0), [voD1] "v"(voD1),
                 [voC] "v"(voC), [voKp] "v"(cur.voKp), [voVp] "v"(cur.voVp), [bK] "s"(cur.bK),
                 [bV] "s"(cur.bV), [bQ] "s"(cur.bQ), [bD] "s"(cur.bD), [bL] "s"(cur.bL), [bT] "s"(cur.bT), [tq] "s"(tq), [tk] "s"(tk),
                 [ldk2] "s"(ldk2), [ldv2] "s"(ldv2), [first] "s"(first), [stQ] "s"(stQ), [stD] "s"(stD), [npair] "s"(npair), [c] "s"(cbits),
                 [lds0] "s"(lds0), [wave] "s"(wv), [pre] "n"(PRE ? 1 : 0)
               : "memory", "vcc", "scc", D4_CLOBBER_A, D4_CLOBBER_V, D4_CLOBBER_S);

#ifdef D4_STAMPS
  const unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
  if (t + wstep < nitems) {
    // ---- prefetch block: once every wave has left the LDS buffers, request the NEXT item's K / V rows (transit area) and
    // first two tiles; they fly while the accumulators of this item are scaled, summed and stored below
    const Item nx = item(t + wstep);
    asm volatile(".set att_pre, 0\n" D4_ASM_MACROS D4_ASM_MACROS2 R"ASM(
      s_barrier
      D4_SRD_INIT
      D4_KVDMA
      D4_STAGE2
    )ASM" D4_ASM_PURGE D4_ASM_PURGE2
                 :
                 : [voQ0] "v"(voQ0), [voQ1] "v"(voQ1), [voD0] "v"(voD0), [voD1] "v"(voD1), [voC] "v"(voC), [voKp] "v"(nx.voKp), [voVp] "v"(nx.voVp), [rb] "v"(rb), [bK] "s"(nx.bK), [bV] "s"(nx.bV), [bQ] "s"(nx.bQ),
                   [bD] "s"(nx.bD), [bL] "s"(nx.bL), [bT] "s"(nx.bT), [tq] "s"(tq), [tk] "s"(tk), [ldk2] "s"(ldk2), [ldv2] "s"(ldv2),
                   [stQ] "s"(stQ), [stD] "s"(stD), [lds0] "s"(lds0), [wave] "s"(wv), [c] "s"(cbits)
                 : "memory", "scc", "v29", "v30", D4_CLOBBER_S, D4_A8(13), D4_A8(14), D4_A8(15), D4_A8(16), D4_A8(17), D4_A8(18), "a128", "a129", "a190", "a191");
  }
#ifdef D4_STAMPS
  const unsigned long long st3b = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: lane (r, h) holds dK / dV [key kw0 + 32 kb + r][d = 32 db + 8 a + 4 h + e] in register 4 a + e of (kb, db).
  // Lanes r and r + 32 hold the two halves of each 8-column group a: one v_permlane32_swap per dword hands lane (r, 0) all of
  // group 2 m and lane (r, 1) all of group 2 m + 1 -> 16-byte stores (4 per tensor and key block instead of 16 8-byte ones)
  auto row16 = [&](const f32x16& acc, int m, float mul) {
    const unsigned x0 = pack2bf(acc[8 * m] * mul, acc[8 * m + 1] * mul), x1 = pack2bf(acc[8 * m + 2] * mul, acc[8 * m + 3] * mul);
    const unsigned y0 = pack2bf(acc[8 * m + 4] * mul, acc[8 * m + 5] * mul), y1 = pack2bf(acc[8 * m + 6] * mul, acc[8 * m + 7] * mul);
    const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false), s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
    return o;
  };
  auto store = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    const int kb0 = kw0 + 32 * kb, ki = kb0 + r;
    f32x16 dk[2], dv[2];
    dk[0] = d4_get16<64 * kb>(); dk[1] = d4_get16<64 * kb + 16>();
    dv[0] = d4_get16<64 * kb + 32>(); dv[1] = d4_get16<64 * kb + 48>();
    unsigned short* dkrow = p.dk + (long)b * p.dk_bs + (long)ki * p.lddk + hd * 64;
    unsigned short* dvrow = p.dv + (long)b * p.dv_bs + (long)ki * p.lddv + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int d = 32 * db + 8 * (2 * m + h);
        const u32x4 pk = row16(dk[db], m, p.ls), pv = row16(dv[db], m, 1.0f);  // (every lane takes part in the swaps)
        if (ki < p.Tk) {
          *(u32x4*)(dkrow + d) = pk;
          *(u32x4*)(dvrow + d) = pv;
        }
      }
    if (p.cs_v && kb0 < p.Tk) {  // v-projection bias gradient: column sums over the block's 32 keys of the bf16 values written.
      // Halving butterfly over the 32 lanes of a half-wave: a lane keeps the half of its values its lane bit selects and adds
      // the partner's copy of that half — 31 exchanges for 32 sums (att_colsum_store: 160), same pairing, same bits.
      float cv[32];
      const bool ok = ki < p.Tk;
#pragma unroll
      for (int i = 0; i < 32; ++i) cv[i] = ok ? bf2f(f2bf(dv[i >> 4][i & 15])) : 0.f;
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        const bool up = (r & m) != 0;
#pragma unroll
        for (int i = 0; i < m; ++i) {
          const float keep = up ? cv[i + m] : cv[i], send = up ? cv[i] : cv[i + m];
          cv[i] = keep + __shfl_xor(send, m, 64);
        }
      }
      // lane r now holds value index r = 16 db + 4 a + e, i.e. column 32 db + 8 a + 4 h + e
      float* dst = p.cs_v + ((long)b * ((p.Tk + 31) >> 5) + (kb0 >> 5)) * (p.H * 64) + hd * 64;
      dst[32 * (r >> 4) + 8 * ((r >> 2) & 3) + 4 * h + (r & 3)] = cv[0];
    }
  };
#ifdef D4_STAMPS
  unsigned long long st1, st2;
  asm volatile("s_mov_b64 %0, s[76:77]\n s_mov_b64 %1, s[78:79]" : "=s"(st1), "=s"(st2));
#endif
  store(IntC<0>{});
  store(IntC<1>{});
#ifdef D4_STAMPS
  const unsigned long long st4 = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    const int o = (t == w0) ? 0 : 8;  // first item of a workgroup | later items
    atomicAdd(&d4_dbg[o + 0], 1ull);
    atomicAdd(&d4_dbg[o + 1], st1 - st0);  // prologue: descriptors, (first item: requests), zeroing, wait, barrier
    atomicAdd(&d4_dbg[o + 2], st2 - st1);  // block 0 + the iteration loop
    atomicAdd(&d4_dbg[o + 3], st3 - st2);  // last dV / dK, drain
    atomicAdd(&d4_dbg[o + 4], st3b - st3);  // prefetch block
    atomicAdd(&d4_dbg[o + 5], st4 - st3b);  // accumulator read-back, column sums, stores (not drained)
  }
  st0 = st4;
#endif
  }  // work items of this workgroup
}

static int attn_fill(const wft_attn_args* a, AttnP& p) {
  p.q = a->q; p.ldq = a->ldq; p.q_bs = a->q_bs;
  p.k = a->k; p.ldk = a->ldk; p.k_bs = a->k_bs;
  p.v = a->v; p.ldv = a->ldv; p.v_bs = a->v_bs;
  p.o = a->o; p.ldo = a->ldo; p.o_bs = a->o_bs;
  p.lse = a->lse;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = a->causal;
  p.scale = a->scale;
  p.qpre = a->q_prescaled ? 1 : 0;
  p.c = p.qpre ? 1.0f : a->scale * LOG2E;
  p.ls = p.qpre ? LN2 : a->scale;
  p.d_o = a->d_o; p.lddo = a->lddo; p.do_bs = a->do_bs;
  p.delta = a->delta;
  p.dq = a->dq; p.lddq = a->lddq; p.dq_bs = a->dq_bs;
  p.dk = a->dk; p.lddk = a->lddk; p.dk_bs = a->dk_bs;
  p.dv = a->dv; p.lddv = a->lddv; p.dv_bs = a->dv_bs;
  p.cs_q = nullptr; p.cs_v = nullptr;
  static const int xcd_on = [] { const char* e = wft_dev_getenv("WFT_ATTN_XCD"); return e ? atoi(e) : 1; }();
  p.xcd = xcd_on;
  return 0;
}

// ------------------------------------------------------------------------------ dQ, one wave per SIMD (round 4)
// The mirror image of attn_bwd_dkdv4w_kernel with the roles of queries and keys exchanged: a wave owns 64 QUERIES (two 32-query
// blocks; Q / dO row fragments in a[64:127] as B operands, dQ^T accumulators a[0:63]) and streams 32-key blocks from LDS tiles
// {K, V} of 64 keys (same piece layout as the Q / dO tiles above).  Scores are formed transposed, S^T[key, query], so a lane holds
// ONE query's column: the row constants -lse/scale and -delta are per-lane values, spread over 16 registers each and used as the C
// operand of the first MFMA of every chain (exactly the 8-wave kernel's arithmetic: results are bit-identical).  Per iteration:
//   MFMA  slots  0-7   dQ^T += K^T(tr) . dS of block j-1      slots 8-23  S^T / dP^T of block j+1 (four chains of four)
//   VALU  slots  0-15  c-multiply of query block 1, both exponentials, dS multiply of query block 0 (24 cycles per slot)
//         slots 16-23  the other dS multiplies, the 16 packs, the c-multiplies of the NEXT block's query block 0
//   LDS   slots  0-7   K / V row fragments of block j+1;  slots 12-19 transposed K fragments of block j;  counted waits only
// Keys past Tk exist only in the last tile: its two iterations are emitted a second time with a per-element select (x -> -1e30
// before the exponential, so p = dS = 0 exactly like the 8-wave kernel's mask); K / V rows past Tk arrive as zeros.
// The kernel also writes the row constants -delta = -rowsum(dO * O) and -lse / scale for the dK/dV kernel (same expression, same
// order as attn_bwd_dq_kernel).
#define Q4_BUF 20480
#define Q4_LDS (3 * Q4_BUF + 4 * D4_KV)  // three {K, V} tile buffers + one Q / dO transit area per wave
#define Q4_ASM_MACROS R"ASM(
; registers: S(g,qb) v[g+32qb..+15], dP(g,qb) v[g+16+32qb..+15], g = 128 / 192; DSF(qb) v[32+8qb..+7]; TK(ks,db) v[48+8ks+4db..+3];
; -delta of the lane's query, 16 copies: v[64+16qb..]; -lse/scale: v[96+16qb..]; dQ^T(qb,db) a[32qb+16db..+15];
; QF(qb,s) a[64+32qb+4s..+3], DOF(qb,s) a[80+32qb+4s..+3]; key-row fragments AK(s) a[128+4s..], AV(s) a[144+4s..]
.macro Q4_M2 n
  v_mfma_f32_32x32x16_bf16 a[32*((\n)/4)+16*((\n)%%2):32*((\n)/4)+16*((\n)%%2)+15], v[48+8*(((\n)/2)%%2)+4*((\n)%%2):48+8*(((\n)/2)%%2)+4*((\n)%%2)+3], v[32+8*((\n)/4)+4*(((\n)/2)%%2):32+8*((\n)/4)+4*(((\n)/2)%%2)+3], a[32*((\n)/4)+16*((\n)%%2):32*((\n)/4)+16*((\n)%%2)+15]
.endm
.macro Q4_M1 n, g
  ; chain (\n)/4: 0 S qb0, 1 dP qb0, 2 S qb1, 3 dP qb1; k-step (\n)%%4
  .if ((\n) %% 4) == 0
    v_mfma_f32_32x32x16_bf16 v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15], a[128+16*(((\n)/4)%%2):128+16*(((\n)/4)%%2)+3], a[64+16*(((\n)/4)%%2)+32*((\n)/8):64+16*(((\n)/4)%%2)+32*((\n)/8)+3], v[96-32*(((\n)/4)%%2)+16*((\n)/8):96-32*(((\n)/4)%%2)+16*((\n)/8)+15]
  .else
    v_mfma_f32_32x32x16_bf16 v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15], a[128+16*(((\n)/4)%%2)+4*((\n)%%4):128+16*(((\n)/4)%%2)+4*((\n)%%4)+3], a[64+16*(((\n)/4)%%2)+32*((\n)/8)+4*((\n)%%4):64+16*(((\n)/4)%%2)+32*((\n)/8)+4*((\n)%%4)+3], v[\g+16*(((\n)/4)%%2)+32*((\n)/8):\g+16*(((\n)/4)%%2)+32*((\n)/8)+15]
  .endif
.endm
; row read i (0..7) of a 32-key block (half \hf of its tile): 0-3 K rows -> AK, 4-7 V rows -> AV
.macro Q4_RD1 i, rb, hf
  .if (\i) < 4
    ds_read_b128 a[128+4*(\i):128+4*(\i)+3], \rb offset:512*\hf+32*(\i)
  .else
    ds_read_b128 a[144+4*((\i)-4):144+4*((\i)-4)+3], \rb offset:10240+512*\hf+32*((\i)-4)
  .endif
.endm
; transposed K read m (0..7) in the order the dQ MFMAs consume them: (ks, db) = (m/4, (m/2)%%2), t = m%%2
.macro Q4_RD2 m, tb, hf
  ds_read_b64_tr_b16 v[48+8*((\m)/4)+4*(((\m)/2)%%2)+2*((\m)%%2):48+8*((\m)/4)+4*(((\m)/2)%%2)+2*((\m)%%2)+1], \tb offset:5152*((\m)%%2)+128*(4*\hf+2*((\m)/4))+64*(((\m)/2)%%2)
.endm
; next 64-key tile: K / V sources += 64 rows, bounds shrink with them (not below zero)
.macro Q4_ADVANCE
  s_add_u32 s40, s40, s56
  s_addc_u32 s41, s41, 0
  s_sub_u32 s42, s42, s56
  s_cselect_b32 s42, 0, s42
  s_add_u32 s44, s44, s57
  s_addc_u32 s45, s45, 0
  s_sub_u32 s46, s46, s57
  s_cselect_b32 s46, 0, s46
.endm
; step i (0..3) of this wave's share of one tile -> the buffer at LDS offset \boff: pieces 2 wave, 2 wave + 1 of K (0, 1) and V (2, 3)
.macro Q4_DMA i, boff
  .if (\i) == 0
    s_add_u32 m0, s58, \boff
    s_nop 0
    buffer_load_dwordx4 %[voK0], s[40:43], 0 offen lds
  .elseif (\i) == 1
    s_add_u32 m0, s59, \boff
    s_nop 0
    buffer_load_dwordx4 %[voK1], s[40:43], 0 offen lds
  .elseif (\i) == 2
    s_add_u32 m0, s58, \boff
    s_add_u32 m0, m0, 10240
    s_nop 0
    buffer_load_dwordx4 %[voV0], s[44:47], 0 offen lds
  .else
    s_add_u32 m0, s59, \boff
    s_add_u32 m0, m0, 10240
    s_nop 0
    buffer_load_dwordx4 %[voV1], s[44:47], 0 offen lds
  .endif
.endm
.macro Q4_STAGE boff
  Q4_DMA 0, \boff
  Q4_DMA 1, \boff
  Q4_DMA 2, \boff
  Q4_DMA 3, \boff
.endm
; one iteration.  gV: generation whose block is exponentiated, gM: target of the S^T / dP^T MFMAs; rb1, hf1: base / tile half of the
; block whose row fragments are read; tb2, hf2: of the block whose transposed fragments are read; dma: LDS-DMA of the tile two
; ahead in slots 8-11; mask: keys at or past \lim + (8 a + e) are switched off (\lim: VGPR = Tk - first key of the block - 4 h)
.macro Q4_ITER gV, gM, rb1, hf1, tb2, hf2, dma, mask, lim
  .set q4_s, 0
  .rept 24
    .if q4_s < 4
      s_waitcnt lgkmcnt(6-q4_s)
    .elseif q4_s == 8
      s_waitcnt lgkmcnt(4)
    .elseif q4_s == 12
      s_waitcnt lgkmcnt(0)
    .endif
    .if q4_s < 8
      Q4_M2 q4_s
    .else
      Q4_M1 q4_s-8, \gM
    .endif
    .if q4_s < 16
      .if \mask
        v_cmp_gt_i32 vcc, \lim, 8*(q4_s/4)+(q4_s%%4)
        v_cndmask_b32 v[\gV+q4_s], v31, v[\gV+q4_s], vcc
      .endif
      v_exp_f32 v[\gV+q4_s], v[\gV+q4_s]
      .if att_pre == 0
      v_mul_f32 v[\gV+32+q4_s], %[c], v[\gV+32+q4_s]
      .endif
      .if q4_s < 8
        Q4_RD1 q4_s, \rb1, \hf1
      .endif
      .if q4_s >= 12
        Q4_RD2 q4_s-12, \tb2, \hf2
      .endif
      .if \mask
        v_cndmask_b32 v[\gV+32+q4_s], v31, v[\gV+32+q4_s], vcc
      .endif
      v_exp_f32 v[\gV+32+q4_s], v[\gV+32+q4_s]
      .if q4_s >= 1
        v_mul_f32 v[\gV+16+q4_s-1], v[\gV+q4_s-1], v[\gV+16+q4_s-1]
      .endif
      .if \dma && q4_s >= 8 && q4_s < 12
        Q4_DMA q4_s-8, s65
      .endif
      .if \dma && q4_s == 12
        Q4_ADVANCE
      .endif
    .else
      .if q4_s == 16
        v_mul_f32 v[\gV+16+15], v[\gV+15], v[\gV+16+15]
      .endif
      v_mul_f32 v[\gV+48+2*(q4_s-16)], v[\gV+32+2*(q4_s-16)], v[\gV+48+2*(q4_s-16)]
      v_mul_f32 v[\gV+48+2*(q4_s-16)+1], v[\gV+32+2*(q4_s-16)+1], v[\gV+48+2*(q4_s-16)+1]
      .if q4_s < 20
        Q4_RD2 q4_s-12, \tb2, \hf2
      .endif
      v_cvt_pk_bf16_f32 v[32+(q4_s-16)], v[\gV+16+2*(q4_s-16)], v[\gV+16+2*(q4_s-16)+1]
      .if att_pre == 0
      v_mul_f32 v[\gM+2*(q4_s-16)], %[c], v[\gM+2*(q4_s-16)]
      v_mul_f32 v[\gM+2*(q4_s-16)+1], %[c], v[\gM+2*(q4_s-16)+1]
      .endif
      v_cvt_pk_bf16_f32 v[40+(q4_s-16)], v[\gV+48+2*(q4_s-16)], v[\gV+48+2*(q4_s-16)+1]
    .endif
    .set q4_s, q4_s+1
  .endr
.endm
; one 64-key tile: boundary + its two iterations (+ the buffer rotation)
.macro Q4_PAIR mask
    ; tile T+1 has landed for every wave, tile T-1's buffer is free -> tile T+2 goes into it
    s_waitcnt vmcnt(0)
    s_barrier
    Q4_ITER 128, 192, v24, 1, v25, 0, 1, \mask, v29
    Q4_ITER 192, 128, v27, 0, v25, 1, 0, \mask, v30
    s_mov_b32 s67, s63
    s_mov_b32 s63, s64
    s_mov_b32 s64, s65
    s_mov_b32 s65, s67
    v_add_u32 v24, s63, %[rb]
    v_add_u32 v25, s63, %[tb]
    v_add_u32 v27, s64, %[rb]
.endm
)ASM"
#define Q4_ASM_PURGE R"ASM(
.purgem Q4_M2
.purgem Q4_M1
.purgem Q4_RD1
.purgem Q4_RD2
.purgem Q4_ADVANCE
.purgem Q4_DMA
.purgem Q4_STAGE
.purgem Q4_ITER
.purgem Q4_PAIR
)ASM"
// a0..a159
#define Q4_CLOBBER_A "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", D4_A8(1), D4_A8(2), D4_A8(3), D4_A8(4), D4_A8(5), D4_A8(6), D4_A8(7), D4_A8(8), D4_A8(9), D4_A8(10), D4_A8(11), D4_A8(12), D4_A8(13), D4_A8(14), D4_A8(15)

// descriptors and requests shared by the main block and the prefetch block of attn_bwd_dq4w_kernel (round 6: persistent, like the dK/dV kernel)
#define Q4_ASM_MACROS2 R"ASM(
; descriptors: K s[40:43], V s[44:47] (tile strides s56, s57), Q s[68:71], dO s[72:75]; this wave's Q / dO transit area s53; LDS-DMA
; destinations inside a tile buffer: piece 2 wave (s58), 2 wave + 1 (s59); buffer offsets: cur (tile T) s63, nxt (T+1) s64, ld (T+2) s65
.macro Q4_SRD_INIT
  s_mov_b64 s[40:41], %[bK]
  s_lshr_b32 s61, %[stK], 6
  s_sub_u32 s62, %[tk], 1
  s_mul_i32 s42, s62, s61
  s_add_u32 s42, s42, 128
  s_mov_b32 s43, 0x20000
  s_mov_b64 s[44:45], %[bV]
  s_lshr_b32 s61, %[stV], 6
  s_mul_i32 s46, s62, s61
  s_add_u32 s46, s46, 128
  s_mov_b32 s47, 0x20000
  s_mov_b32 s56, %[stK]
  s_mov_b32 s57, %[stV]
  s_mov_b64 s[68:69], %[bQ]
  s_sub_u32 s62, %[tq], 1
  s_mul_i32 s70, s62, %[ldq2]
  s_add_u32 s70, s70, 128
  s_mov_b32 s71, 0x20000
  s_mov_b64 s[72:73], %[bD]
  s_mul_i32 s74, s62, %[ldd2]
  s_add_u32 s74, s74, 128
  s_mov_b32 s75, 0x20000
  s_mul_i32 s53, %[wave], )ASM" D4_STR(D4_KV) R"ASM(
  s_add_u32 s53, s53, 3*)ASM" D4_STR(Q4_BUF) R"ASM(
  s_add_u32 s53, s53, %[lds0]
  s_lshl_b32 s61, %[wave], 1
  s_mul_i32 s58, s61, 1280
  s_lshr_b32 s62, s61, 1
  s_lshl_b32 s62, s62, 4
  s_add_u32 s58, s58, s62
  s_add_u32 s58, s58, %[lds0]
  s_add_u32 s59, s58, 1344
  s_mov_b32 s63, 0
  s_mov_b32 s64, )ASM" D4_STR(Q4_BUF) R"ASM(
  s_mov_b32 s65, 2*)ASM" D4_STR(Q4_BUF) R"ASM(
.endm
; Q / dO rows of this wave's 64 queries -> its transit area as two tiles in the piece layout (see D4_KVDMA: whole 128-byte rows by
; LDS-DMA instead of row-per-lane fragment loads); read into a[64:127] by the main block
.macro Q4_TRANSIT
  .set q4_i, 0
  .rept 8
    s_mul_i32 s61, %[ldq2], (q4_i%%2)+4*(q4_i/2)
    v_add_u32 v29, s61, %[voQ]
    s_add_u32 m0, s53, q4_i*1280+64*(q4_i%%2)+16*(q4_i/2)
    s_mul_i32 s62, %[ldd2], (q4_i%%2)+4*(q4_i/2)
    buffer_load_dwordx4 v29, s[68:71], 0 offen lds
    v_add_u32 v30, s62, %[voD]
    s_add_u32 m0, s53, 10240+q4_i*1280+64*(q4_i%%2)+16*(q4_i/2)
    s_nop 0
    buffer_load_dwordx4 v30, s[72:75], 0 offen lds
    .set q4_i, q4_i+1
  .endr
.endm
; key tiles 0, 1 -> buffers 0, 1 (sources end up two tiles on)
.macro Q4_STAGE2
  Q4_STAGE s63
  Q4_ADVANCE
  s_nop 4
  Q4_STAGE s64
  Q4_ADVANCE
.endm
)ASM"
#define Q4_ASM_PURGE2 R"ASM(
.purgem Q4_SRD_INIT
.purgem Q4_TRANSIT
.purgem Q4_STAGE2
)ASM"

// Round 6: PERSISTENT like attn_bwd_dkdv4w_kernel — one workgroup per CU (it owns the CU: 143 KB of LDS, one wave per SIMD) walks work
// items (batch, head, 256-query block) in XCD-local order.  One workgroup per item paid every item's cold start in full (LDS is not
// shared between two of them, so nothing overlapped: the dK/dV kernel's stamps put a first item's prologue at 11.6 k ticks against 4.1 k
// for a later one): here the NEXT item's Q / dO rows, its first two key tiles and its row constants (-delta, -lse / ls: the O / dO / lse
// loads) are requested behind this item's loop and fly while its accumulators are scaled, summed and stored.
template <bool PRE>
__global__ __launch_bounds__(256) void attn_bwd_dq4w_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nqb = (p.Tq + 255) >> 8;
  const int ngrp = p.H * p.B;
  const bool xcd_mode = ((ngrp & 7) == 0) && p.xcd && ((gridDim.x & 7) == 0);
  const int xcd = xcd_mode ? (int)(blockIdx.x & 7) : 0;
  const int w0 = xcd_mode ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wstep = xcd_mode ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int nitems = (xcd_mode ? ngrp >> 3 : ngrp) * nqb;
  if (w0 >= nitems) return;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // fragment read bases in a tile buffer (layout: see attn_bwd_dkdv4w_kernel)
  const int c = r & 15, pidr = (c & 1) | ((c >> 2) << 1);
  const unsigned rb = lds0 + pidr * D4_PIECE + 64 * (pidr & 1) + 16 * (pidr >> 1) + (2 * (r >> 4) + ((c >> 1) & 1)) * 128 + h * 16;
  const int g4 = lane >> 4, i16 = lane & 15, pidt = ((i16 >> 2) & 1) | ((g4 >> 1) << 1);
  const unsigned tb = lds0 + pidt * D4_PIECE + 64 * (pidt & 1) + 16 * (pidt >> 1) + ((i16 >> 3) & 1) * 128 + 32 * (g4 & 1) + 8 * (i16 & 3);
  // LDS-DMA share of this wave: pieces 2 wave and 2 wave + 1 of K and of V
  const int slot = lane >> 3, ch = lane & 7;
  auto krow = [&](int pid) { return (pid & 1) + 4 * (pid >> 1) + 2 * (slot & 1) + 16 * (slot >> 1); };
  const unsigned voK0 = (unsigned)(krow(2 * wave) * (int)p.ldk + ch * 8) * 2u, voK1 = (unsigned)(krow(2 * wave + 1) * (int)p.ldk + ch * 8) * 2u;
  const unsigned voV0 = (unsigned)(krow(2 * wave) * (int)p.ldv + ch * 8) * 2u, voV1 = (unsigned)(krow(2 * wave + 1) * (int)p.ldv + ch * 8) * 2u;
  auto sg64 = [](unsigned long long x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const unsigned tq = __builtin_amdgcn_readfirstlane((unsigned)p.Tq), tk = __builtin_amdgcn_readfirstlane((unsigned)p.Tk);
  const unsigned ldq2 = __builtin_amdgcn_readfirstlane((unsigned)p.ldq * 2u), ldd2 = __builtin_amdgcn_readfirstlane((unsigned)p.lddo * 2u);
  const unsigned stK = __builtin_amdgcn_readfirstlane((unsigned)p.ldk * 128u), stV = __builtin_amdgcn_readfirstlane((unsigned)p.ldv * 128u);
  const unsigned npair = __builtin_amdgcn_readfirstlane((unsigned)((p.Tk + 63) >> 6));  // 64-key tiles
  const int lim0 = p.Tk - 64 * ((p.Tk + 63) / 64 - 1) - 4 * h;  // keys left from the first key of the LAST tile, minus 4 h
  const float cscale = p.c;
  const unsigned cbits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, cscale));
  const unsigned wv = (unsigned)wave;

  // everything that depends on the work item
  struct Item {
    int b, hd, qw0;
    unsigned long long bQ, bD, bK, bV;
    unsigned voQ, voD;
    float nl[2], nd[2];
  };
  auto coords = [&](int t, Item& x) {
    const int g = xcd_mode ? (t / nqb) * 8 + xcd : t / nqb;
    x.hd = g % p.H;
    x.b = g / p.H;
    x.qw0 = (t % nqb) * 256 + wave * 64;
    x.bQ = sg64((unsigned long long)(p.q + (long)x.b * p.q_bs + x.hd * 64));
    x.bD = sg64((unsigned long long)(p.d_o + (long)x.b * p.do_bs + x.hd * 64));
    x.bK = sg64((unsigned long long)(p.k + (long)x.b * p.k_bs + x.hd * 64));
    x.bV = sg64((unsigned long long)(p.v + (long)x.b * p.v_bs + x.hd * 64));
    // this lane's share of Q / dO piece 0 of the wave's 64 queries (slot: query 2 (slot & 1) + 16 (slot >> 1), chunk lane & 7): byte
    // offsets relative to the (batch, head) bases (rows past Tq load zeros)
    x.voQ = (unsigned)((x.qw0 + 2 * (slot & 1) + 16 * (slot >> 1)) * (int)p.ldq + ch * 8) * 2u;
    x.voD = (unsigned)((x.qw0 + 2 * (slot & 1) + 16 * (slot >> 1)) * (int)p.lddo + ch * 8) * 2u;
  };
  // row constants of this lane's two queries (query blocks 0 / 1 of the wave), negated; written for the dK/dV kernel
  auto row_consts = [&](Item& x) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = x.qw0 + 32 * qb + r;
      const int qc = qi < p.Tq ? qi : p.Tq - 1;
      const long sidx = ((long)x.b * p.H + x.hd) * p.Tq + qc;
      x.nl[qb] = -p.lse[sidx] / p.ls;
      const unsigned short* orow = p.o + (long)x.b * p.o_bs + (long)qc * p.ldo + x.hd * 64;
      const unsigned short* dorow = p.d_o + (long)x.b * p.do_bs + (long)qc * p.lddo + x.hd * 64;
      float part = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 of = att_load_reg_frag(orow, s, h), df = att_load_reg_frag(dorow, s, h);
        const u32x4 ou = __builtin_bit_cast(u32x4, of), du = __builtin_bit_cast(u32x4, df);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          part += bf2f((unsigned short)(ou[e] & 0xffff)) * bf2f((unsigned short)(du[e] & 0xffff));
          part += bf2f((unsigned short)(ou[e] >> 16)) * bf2f((unsigned short)(du[e] >> 16));
        }
      }
      x.nd[qb] = -(part + __shfl_xor(part, 32, 64));
      if (h == 0 && qi < p.Tq) {
        p.delta[sidx] = x.nd[qb];
        p.delta[(long)p.B * p.H * p.Tq + sidx] = x.nl[qb];
      }
    }
  };
  Item cur;
  coords(w0, cur);
  row_consts(cur);

  for (int t = w0; t < nitems; t += wstep) {
  const int b = cur.b, hd = cur.hd, qw0 = cur.qw0;
  const unsigned first = __builtin_amdgcn_readfirstlane((unsigned)(t == w0));
  asm volatile(".set att_pre, %c[pre]\n" Q4_ASM_MACROS Q4_ASM_MACROS2 R"ASM(
    Q4_SRD_INIT
    s_sub_u32 s66, %[npair], 1      ; loop counter
    s_cmp_eq_u32 %[first], 0
    s_cbranch_scc1 2f
    ; ---- first item of this workgroup: its Q / dO rows and key tiles 0, 1 are requested here ...
    Q4_TRANSIT
    Q4_STAGE2
    s_branch 3f
2:
    ; ---- ... later ones found them requested by the prefetch block behind the previous item (below): only the descriptors move on
    Q4_ADVANCE
    Q4_ADVANCE
    s_waitcnt vmcnt(0)
3:
    ; ---- (under the loads) accumulators, packed dS and transposed fragments start from zero; the row constants spread out
    .set q4_i, 0
    .rept 64
      v_accvgpr_write_b32 a[q4_i], 0
      .set q4_i, q4_i+1
    .endr
    .set q4_i, 32
    .rept 32
      v_mov_b32 v[q4_i], 0
      .set q4_i, q4_i+1
    .endr
    .set q4_i, 0
    .rept 16
      v_mov_b32 v[64+q4_i], %[nd0]
      v_mov_b32 v[80+q4_i], %[nd1]
      v_mov_b32 v[96+q4_i], %[nl0]
      v_mov_b32 v[112+q4_i], %[nl1]
      .set q4_i, q4_i+1
    .endr
    v_mov_b32 v31, 0xf149f2ca      ; -1e30
    v_mov_b32 v29, %[lim0]
    v_add_u32 v30, -32, v29
    v_mov_b32 v24, %[rb]
    v_mov_b32 v25, %[tb]
    v_add_u32 v27, s64, v24
    s_waitcnt vmcnt(4)             ; Q / dO rows and tile 0 (tile 1: four pieces may still be in flight)
    s_barrier
    ; Q / dO row fragments (lane (r, h): query 32 qb + r, columns 16 s + 8 h .. + 8) from the transit area -> a[64:127]
    s_sub_u32 s61, s53, %[lds0]
    v_add_u32 v26, s61, %[rb]
    .set q4_i, 0
    .rept 8
      ds_read_b128 a[64+32*(q4_i/4)+4*(q4_i%%4):64+32*(q4_i/4)+4*(q4_i%%4)+3], v26 offset:512*(q4_i/4)+32*(q4_i%%4)
      ds_read_b128 a[80+32*(q4_i/4)+4*(q4_i%%4):80+32*(q4_i/4)+4*(q4_i%%4)+3], v26 offset:10240+512*(q4_i/4)+32*(q4_i%%4)
      .set q4_i, q4_i+1
    .endr
    ; ---- block 0: row fragments, S^T / dP^T -> generation 128
    .set q4_i, 0
    .rept 8
      Q4_RD1 q4_i, v24, 0
      .set q4_i, q4_i+1
    .endr
    s_waitcnt lgkmcnt(0)
    .set q4_i, 0
    .rept 16
      Q4_M1 q4_i, 128
      .set q4_i, q4_i+1
    .endr
    s_nop 15
    s_nop 15
    .set q4_i, 0
    .if att_pre == 0
    .rept 16
      v_mul_f32 v[128+q4_i], %[c], v[128+q4_i]
      .set q4_i, q4_i+1
    .endr
    .endif
    s_cmp_eq_u32 s66, 0
    s_cbranch_scc1 5f
1:
    Q4_PAIR 0
    s_sub_u32 s66, s66, 1
    s_cmp_eq_u32 s66, 0
    s_cbranch_scc0 1b
5:
    ; ---- the last tile: keys past Tk are switched off element by element
    Q4_PAIR 1
    ; ---- dQ of the last block
    s_waitcnt lgkmcnt(0)
    s_nop 1
    .set q4_i, 0
    .rept 8
      Q4_M2 q4_i
      .set q4_i, q4_i+1
    .endr
    s_waitcnt vmcnt(0)
    s_nop 15
  )ASM" Q4_ASM_PURGE Q4_ASM_PURGE2
               :
               : [rb] "v"(rb), [tb] "v"(tb), [voK0] "v"(voK0), [voK1] "v"(voK1), [voV0] "v"(voV0), [voV1] "v"(voV1), [voQ] "v"(cur.voQ),
                 [voD] "v"(cur.voD), [nl0] "v"(cur.nl[0]), [nl1] "v"(cur.nl[1]), [nd0] "v"(cur.nd[0]), [nd1] "v"(cur.nd[1]), [lim0] "v"(lim0),
                 [bK] "s"(sg64(cur.bK)), [bV] "s"(sg64(cur.bV)), [bQ] "s"(sg64(cur.bQ)), [bD] "s"(sg64(cur.bD)), [tq] "s"(tq), [tk] "s"(tk), [ldq2] "s"(ldq2),
                 [ldd2] "s"(ldd2), [stK] "s"(stK), [stV] "s"(stV), [npair] "s"(npair), [c] "s"(cbits), [lds0] "s"(lds0), [wave] "s"(wv),
                 [first] "s"(first), [pre] "n"(PRE ? 1 : 0)
               : "memory", "vcc", "scc", Q4_CLOBBER_A, D4_CLOBBER_V, "v30", "v31", D4_CLOBBER_S);

  Item nx;
  const bool more = t + wstep < nitems;
  if (more) {
    // ---- prefetch block: once every wave has left the tile buffers, request the NEXT item's Q / dO rows (transit areas) and first
    // two key tiles, then form its row constants: all of it flies while this item's accumulators are scaled, summed and stored below
    coords(t + wstep, nx);
    asm volatile(".set att_pre, 0\n" Q4_ASM_MACROS Q4_ASM_MACROS2 R"ASM(
      s_barrier
      Q4_SRD_INIT
      Q4_TRANSIT
      Q4_STAGE2
    )ASM" Q4_ASM_PURGE Q4_ASM_PURGE2
                 :
                 : [rb] "v"(rb), [tb] "v"(tb), [c] "s"(cbits), [voK0] "v"(voK0), [voK1] "v"(voK1), [voV0] "v"(voV0), [voV1] "v"(voV1), [voQ] "v"(nx.voQ), [voD] "v"(nx.voD), [bK] "s"(sg64(nx.bK)),
                   [bV] "s"(sg64(nx.bV)), [bQ] "s"(sg64(nx.bQ)), [bD] "s"(sg64(nx.bD)), [tq] "s"(tq), [tk] "s"(tk), [ldq2] "s"(ldq2), [ldd2] "s"(ldd2),
                   [stK] "s"(stK), [stV] "s"(stV), [lds0] "s"(lds0), [wave] "s"(wv)
                 : "memory", "scc", "v29", "v30", D4_CLOBBER_S);
    row_consts(nx);
  }

  // ---- epilogue: lane (r, h) holds dQ [query qw0 + 32 qb + r][d = 32 db + 8 a + 4 h + e] in register 4 a + e of (qb, db)
  auto row16 = [&](const f32x16& acc, int m, float mul) {
    const unsigned x0 = pack2bf(acc[8 * m] * mul, acc[8 * m + 1] * mul), x1 = pack2bf(acc[8 * m + 2] * mul, acc[8 * m + 3] * mul);
    const unsigned y0 = pack2bf(acc[8 * m + 4] * mul, acc[8 * m + 5] * mul), y1 = pack2bf(acc[8 * m + 6] * mul, acc[8 * m + 7] * mul);
    const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false), s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
    return o;
  };
  auto store = [&](auto qbc) {
    constexpr int qb = decltype(qbc)::value;
    const int qb0 = qw0 + 32 * qb, qi = qb0 + r;
    f32x16 dq[2];
    dq[0] = d4_get16<32 * qb>(); dq[1] = d4_get16<32 * qb + 16>();
    unsigned short* drow = p.dq + (long)b * p.dq_bs + (long)qi * p.lddq + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const u32x4 pk = row16(dq[db], m, p.scale);  // (every lane takes part in the swaps)
        if (qi < p.Tq) *(u32x4*)(drow + 32 * db + 8 * (2 * m + h)) = pk;
      }
    if (p.cs_q && qb0 < p.Tq) {  // q-projection bias gradient: column sums of the bf16 values written (halving butterfly, see dK/dV)
      float cv[32];
      const bool ok = qi < p.Tq;
#pragma unroll
      for (int i = 0; i < 32; ++i) cv[i] = ok ? bf2f(f2bf(dq[i >> 4][i & 15] * p.scale)) : 0.f;
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        const bool up = (r & m) != 0;
#pragma unroll
        for (int i = 0; i < m; ++i) {
          const float keep = up ? cv[i + m] : cv[i], send = up ? cv[i] : cv[i + m];
          cv[i] = keep + __shfl_xor(send, m, 64);
        }
      }
      float* dst = p.cs_q + ((long)b * ((p.Tq + 31) >> 5) + (qb0 >> 5)) * (p.H * 64) + hd * 64;
      dst[32 * (r >> 4) + 8 * ((r >> 2) & 3) + 4 * h + (r & 3)] = cv[0];
    }
  };
  store(IntC<0>{});
  store(IntC<1>{});
  if (more) cur = nx;
  }  // work items of this workgroup
}


#define ATT_ALIGNED(ptr, ld, bs) ((((uintptr_t)(ptr)) & 15) == 0 && ((ld) % 8) == 0 && ((bs) % 8) == 0)

// Forward kernel choice: 0 (default) = attn_fwd_pipe_kernel (software-pipelined: S of tile kt+1 behind the softmax of tile kt) for
// non-causal calls with Tk >= 512 — the encoder — and attn_fwd_kernel for the rest (the short key ranges of the decoder: the
// four-slot ring's longer prologue costs 2-5 % there); 1 = attn_fwd_kernel everywhere.  Bit-identical results either way.
// (Round 4's one-wave-per-SIMD forward kernel measured equal to attn_fwd_kernel and was removed in round 5.)
static int g_fwd_variant = [] { const char* e = wft_dev_getenv("WFT_FWD_VARIANT"); return (e && !strcmp(e, "8w")) ? 1 : 0; }();
static bool wft_fwd_pipe_eligible(const wft_attn_args* a) {
  static const int min_tk = [] { const char* e = wft_dev_getenv("WFT_FWDPIPE_MIN_TK"); return e ? atoi(e) : 512; }();
  return g_fwd_variant == 0 && !(a->variant & 1) && !a->causal && a->Tk >= min_tk;
}

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
  if (wft_fwd_pipe_eligible(a)) hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, block, 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, (hipStream_t)stream, p);
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

// Which dK/dV kernel: 0 (default) the one-wave-per-SIMD kernel where it applies, 1 always the 8-wave kernel.  WFT_DKDV_VARIANT=8w|4w
// sets the start value; returns the previous one (a negative argument only reads).
// persistent dK/dV launches (one workgroup per CU) or one item per workgroup: WFT_ATTN_PERSISTENT=0 at load time (engine/lib.py sets it
// in a multi-GPU job), wft_attn_set_persistent() inside a process (bench.py's ddp_mode_1gpu block)
static int g_attn_persistent = [] { const char* e = getenv("WFT_ATTN_PERSISTENT"); return (e && e[0] == '0') ? 0 : 1; }();
static int g_dkdv_variant = [] { const char* e = wft_dev_getenv("WFT_DKDV_VARIANT"); return (e && !strcmp(e, "8w")) ? 1 : 0; }();
// the same for the dQ kernel (attn_bwd_dq4w_kernel): WFT_DQ_VARIANT=8w|4w
static int g_dq_variant = [] { const char* e = wft_dev_getenv("WFT_DQ_VARIANT"); return (e && !strcmp(e, "8w")) ? 1 : 0; }();
// non-causal calls with enough queries to fill 256-query workgroups; byte offsets must fit the asm block's 32-bit buffer addressing
static bool wft_dq4w_eligible(const wft_attn_args* a) {
  static const int min_tq = [] { const char* e = wft_dev_getenv("WFT_DQ4W_MIN_TQ"); return e ? atoi(e) : 512; }();
  if (g_dq_variant != 0 || (a->variant & 2) || a->causal || a->Tq < min_tq) return false;
  const long lim = 0x7fffffffL;
  return (long)(a->Tq + 256) * a->ldq * 2 < lim && (long)(a->Tq + 256) * a->lddo * 2 < lim && (long)(a->Tk + 256) * a->ldk * 2 < lim &&
         (long)(a->Tk + 256) * a->ldv * 2 < lim;
}
// non-causal sweeps over at least two 64-query tiles whose byte offsets fit the 32-bit buffer addressing of the asm block
static bool wft_dkdv4w_eligible(const wft_attn_args* a) {
  if (g_dkdv_variant != 0 || (a->variant & 4) || a->causal || a->Tq < 128) return false;
  const long lim = 0x7fffffffL;
  // (+ 256: the last workgroup's lanes address rows up to 255 past the end; the descriptors return zeros for them)
  return (long)(a->Tq + 256) * a->ldq * 2 < lim && (long)(a->Tq + 256) * a->lddo * 2 < lim && (long)(a->Tk + 256) * a->ldk * 2 < lim &&
         (long)(a->Tk + 256) * a->ldv * 2 < lim;
}

// Which kernel serves these arguments (pure host function; bench.py / tests attribute timings and assert the dispatch):
// which = 0 forward: 2 attn_fwd_pipe_kernel, 1 attn_fwd_kernel; 1 dQ: 4 attn_bwd_dq4w_kernel, 8 attn_bwd_dq_kernel;
// 2 dK/dV: 4 attn_bwd_dkdv4w_kernel, 8 attn_bwd_dkdv_kernel
extern "C" int wft_attn_variant(const wft_attn_args* a, int which) {
  if (!a) return WFT_ERR_ARG;
  if (which == 0) return wft_fwd_pipe_eligible(a) ? 2 : 1;
  if (which == 1) return wft_dq4w_eligible(a) ? 4 : 8;
  if (which == 2) return wft_dkdv4w_eligible(a) ? 4 : 8;
  return WFT_ERR_ARG;
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
  if (wft_dq4w_eligible(a)) {
    static bool ldsq_set[64] = {false};
    int devq = 0;
    if (hipGetDevice(&devq) != hipSuccess || devq < 0 || devq >= 64) devq = 0;
    if (!ldsq_set[devq]) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq4w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_LDS);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dq4w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_LDS);
      if (e != hipSuccess) {
        wft_set_error("wft_attn_bwd_bf16: the dQ kernel needs %d bytes of dynamic LDS, hipFuncSetAttribute: %s", Q4_LDS, hipGetErrorString(e));
        return WFT_ERR_LAUNCH;
      }
      ldsq_set[devq] = true;
    }
    // persistent like the dK/dV kernel below: one workgroup per CU walks the (batch, head, 256-query block) items; launch_mode = 1
    // (or WFT_ATTN_PERSISTENT=0): one item per workgroup
    const long qitems = (long)((a->Tq + 255) / 256) * a->H * a->B;
    long qwgs = (g_attn_persistent != 0 && a->launch_mode != 1) ? wft_num_cus() : qitems;
    if (qwgs > qitems) qwgs = qitems;
    if (((long)a->H * a->B) % 8 == 0 && qwgs >= 8) qwgs -= qwgs % 8;  // XCD mode needs the same number of workgroups on every XCD
    const dim3 gq((unsigned)qwgs);
    if (p.qpre) hipLaunchKernelGGL(attn_bwd_dq4w_kernel<true>, gq, dim3(256), Q4_LDS, s, p);
    else hipLaunchKernelGGL(attn_bwd_dq4w_kernel<false>, gq, dim3(256), Q4_LDS, s, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(((a->Tq + 127) / 128) * a->H * a->B)), dim3(256), 0, s, p);
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  {
    static bool lds_set[64] = {false};  // hipFuncSetAttribute is per device
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
  if (wft_dkdv4w_eligible(a)) {
    static bool lds4_set[64] = {false};
    if (!lds4_set[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkdv4w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dkdv4w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS);
      if (e != hipSuccess) {
        wft_set_error("wft_attn_bwd_bf16: the dK/dV kernel needs %d bytes of dynamic LDS, hipFuncSetAttribute: %s", D4_LDS, hipGetErrorString(e));
        return WFT_ERR_LAUNCH;
      }
      lds4_set[dev] = true;
    }
    // persistent: one workgroup per CU (WFT_DKDV_WGS overrides: A/B runs; >= the number of items = one item per workgroup).
    // WFT_ATTN_PERSISTENT=0 (set by engine/lib.py in a multi-GPU job, like WFT_NT256_PERSISTENT): one item per workgroup — RCCL's
    // collective kernels hold CUs during the backward pass, and a static walk would leave those CUs' share of the items for a
    // second round; the hardware dispatcher balances single-item workgroups (measured equal on one GPU: 732 vs 735 us)
    static const int wgs_env = [] { const char* e = wft_dev_getenv("WFT_DKDV_WGS"); return e ? atoi(e) : 0; }();
    const bool persistent = g_attn_persistent != 0 && a->launch_mode != 1;
    const long items = (long)((a->Tk + 255) / 256) * a->H * a->B;
    long wgs = wgs_env > 0 ? wgs_env : (persistent ? wft_num_cus() : items);
    if (wgs > items) wgs = items;
    if (((long)a->H * a->B) % 8 == 0 && wgs >= 8) wgs -= wgs % 8;  // XCD mode needs the same number of workgroups on every XCD
    if (p.qpre) hipLaunchKernelGGL(attn_bwd_dkdv4w_kernel<true>, dim3((unsigned)wgs), dim3(256), D4_LDS, s, p);
    else hipLaunchKernelGGL(attn_bwd_dkdv4w_kernel<false>, dim3((unsigned)wgs), dim3(256), D4_LDS, s, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((unsigned)(((a->Tk + 127) / 128) * a->H * a->B)), dim3(256), 2 * DKDV_BUF, s, p);
  }
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

