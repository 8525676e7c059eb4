/*
 * wft.h — C ABI of libwft.so, the MI355X (gfx950) kernel library behind the
 * whisper_finetune training hot path.
 *
 * The reference (i4Ds/whisper-finetune) has no FFI of its own: its hot path is
 * Python calling PyTorch / openai-whisper / torchaudio ops.  Each entry point
 * below names the reference call site (path relative to the reference root,
 * file:line) whose arithmetic it replaces.  The Python host side
 * (whisper-finetune_amd/whisper_finetune/engine/lib.py) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    the name ends in _host; the caller (PyTorch's allocator) owns every buffer
 *    including workspaces;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*):
 *    no host sync, no allocation, no default-stream use, no device state and no
 *    mutable host state kept in the library — re-entrant across streams and
 *    threads.  Everything that selects a launch mode or a kernel variant is a
 *    FIELD of the call's argument struct (wft_gemm_args / wft_attn_args:
 *    launch_mode, variant, q_prescaled); the only process-wide inputs are read
 *    ONCE when the library is loaded: WFT_NT256_PERSISTENT / WFT_ATTN_PERSISTENT
 *    (=0: one workgroup per tile everywhere) and, in libwft_timing.so only, the
 *    developer switches of DESIGN.md §3;
 *  - return value: WFT_OK (0) or a negative wft_status; wft_last_error() gives
 *    a thread-local message for the last failure on the calling thread;
 *  - matrices are row-major; `ld*` are leading dimensions in ELEMENTS;
 *  - "bf16" buffers hold IEEE bfloat16 (uint16_t storage), "f32" IEEE float;
 *  - RNG-dependent ops take ALREADY-DRAWN parameters so the host keeps the
 *    reference's CPU-generator draw order (SURVEY.md §7 "RNG parity").
 */
#ifndef WFT_H
#define WFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  WFT_OK = 0,
  WFT_ERR_ARG = -1,     /* bad shape / alignment / null pointer            */
  WFT_ERR_LAUNCH = -2,  /* hipLaunchKernel failed (message has the reason) */
  WFT_ERR_UNSUPPORTED = -3
} wft_status;

typedef uint16_t wft_bf16;

const char* wft_last_error(void);
/* library version + compiled arch string, e.g. "wft 0.1 gfx950" */
const char* wft_version(void);

/* ------------------------------------------------------------------ casts */
/* fp32 master weight -> bf16 shadow (what `W.to(x.dtype)` does on every
 * whisper.model.Linear.forward under autocast).        */
int wft_cast_f32_bf16(const float* src, wft_bf16* dst, int64_t n, void* stream);
int wft_cast_bf16_f32(const wft_bf16* src, float* dst, int64_t n, void* stream);
/* src f32 [rows, cols] -> dst bf16 [rows_pad, cols_pad] (zero padded, row
 * stride ld_dst) and, if dst_t != NULL, dst_t bf16 [cols_pad, rows_pad] (row
 * stride ld_dst_t: the transposed shadow the backward-data GEMM consumes).
 * The strides let several weights (q/k/v) share one concatenated shadow.
 * fwd_scale (0 is read as 1): dst = bf16(fwd_scale * src) — one rounding of the
 * scaled value — while dst_t stays bf16(src): the softmax scale * log2(e) of a
 * self-attention q projection folded into its FORWARD shadow
 * (wft_attn_args.q_prescaled; whisper's `q * scale`, App. A.1).               */
int wft_cast_pad_transpose_f32_bf16(const float* src, int64_t rows, int64_t cols,
                                    wft_bf16* dst, wft_bf16* dst_t,
                                    int64_t rows_pad, int64_t cols_pad,
                                    int64_t ld_dst, int64_t ld_dst_t, float fwd_scale, void* stream);
/* W_eff = W + scaling * B @ (A * mask): the weight minLoRA's parametrization produces for every adapted Linear
 * (reference call sites model/lora.py:30-71 apply, :83-89 merge).  W f32 [rows, cols], B f32
 * [rows, rank], A f32 [rank, cols], mask f32 [cols] or NULL (the already-drawn dropout mask), rank <= 64.
 * Outputs (any subset): dst bf16 [rows_pad, cols_pad] (ld_dst) and dst_t bf16 [cols_pad, rows_pad] (ld_dst_t), zero
 * padded — the GEMM shadows of the training forward/backward; dst_f32 f32 [rows, cols] (may alias W) — merge_lora.
 * fwd_scale: as in wft_cast_pad_transpose_f32_bf16 (dst only; dst_t and dst_f32 are never scaled).                  */
int wft_lora_merge(const float* W, int64_t rows, int64_t cols, const float* B, const float* A, const float* mask,
                   int rank, float scaling, wft_bf16* dst, wft_bf16* dst_t, int64_t rows_pad, int64_t cols_pad,
                   int64_t ld_dst, int64_t ld_dst_t, float* dst_f32, float fwd_scale, void* stream);
/* Operands of the rank-r adapter-gradient GEMMs (du = dy (sB), u = x (s A*mask)^T; minLoRA parametrization, model/lora.py:30-71)
 * of ONE adapter, written into its Linear group's zero-initialised padded bf16 buffers: Am [rpad, K] rows ro..ro+rank (and
 * AmT [K, rpad]) = scaling * A * mask; Bb [npad, rpad] block (no..no+n, ro..ro+rank) (and BbT [rpad, npad]) = scaling * B.
 * A f32 [rank, K], B f32 [n, rank], mask f32 [K] or NULL.                                                                */
int wft_lora_pack(const float* A, const float* mask, const float* B, int rank, int64_t K, int64_t n, float scaling,
                  wft_bf16* Am, wft_bf16* AmT, wft_bf16* Bb, wft_bf16* BbT, int64_t rpad, int64_t npad, int64_t ro,
                  int64_t no, void* stream);
/* wft_lora_merge (dst + dst_t) and wft_lora_pack of EVERY adapter of a model in one launch — issued once per training forward,
 * right after the dropout masks are drawn, instead of two launches per Linear (2 x 512 for large-v3).  tab: device array of n
 * rows x 20 int64:
 *   0 W, 1 rows, 2 cols, 3 B, 4 A, 5 mask (or 0), 6 rank, 7 scaling (the float's bits in bits 0..31; bits 32..63: fwd_scale as float bits, 0 = 1 —
 *   dst = bf16(fwd_scale * w), dst_t unscaled, as in wft_cast_pad_transpose_f32_bf16), 8 dst, 9 dst_t (or 0), 10 ld_dst,
 *   11 ld_dst_t, 12 Am, 13 AmT, 14 Bb, 15 BbT (12-15 all 0: no pack), 16 rpad, 17 npad, 18 ro, 19 no
 * with the meanings of the two per-adapter entry points; rows % 64 == 0, cols % 64 == 0, rank <= 64, W 16-byte and dst / dst_t
 * 8-byte aligned, ld_dst % 4 == 0, ld_dst_t % 4 == 0 (the caller checks: a model that does not fit keeps the per-adapter calls).
 * tile_start: device int32 [n + 1], prefix sums of (rows / 64) * (cols / 64); total_tiles = tile_start[n].  Values are
 * bit-identical to the per-adapter entry points. */
int wft_lora_refresh_mt(const void* tab, const int32_t* tile_start, int n, int total_tiles, void* stream);
/* n small f32 vector copies in one launch.  tab: DEVICE int64 [n][3] = source address, destination address, element count (bits 0..31;
 * bits 32..63: a scale as float bits, 0 = plain copy — the q slice of a fused q/k/v bias under wft_attn_args.q_prescaled)
 * (disjoint ranges).  The host mirror uses it to restack the bias vectors of every fused q/k/v group after an optimizer step
 * (whisper.model.Linear biases: the GEMM epilogue reads ONE bias vector per fused group). */
int wft_mt_copy_f32(const void* tab, int n, void* stream);
/* y[i] = a[i] + b[i] (bf16) — gradient accumulation on the residual stream. */
int wft_add_bf16(const wft_bf16* a, const wft_bf16* b, wft_bf16* y, int64_t n, void* stream);
/* out = a*x + b*y over n bf16 elements (y may be NULL).  StochasticDepthMixin's train-time rescale
 * x + (block(x) - x)/(1-p) (model/model_utils.py:241-250) and its backward in one pass each.   */
int wft_axpby_bf16(float a, const wft_bf16* x, float b, const wft_bf16* y, wft_bf16* out, int64_t n, void* stream);
/* out[i] = dy[i] * gelu'(pre[i]) (exact-erf GELU) — backward through F.gelu in the
 * conv stem (model/model_utils.py:276-277).                                  */
int wft_dgelu_mul_bf16(const wft_bf16* dy, const wft_bf16* pre, wft_bf16* out, int64_t n, void* stream);
/* out[c] (+)= sum_r x[r, c] — bias gradients.  x bf16 [rows, ld], out f32[cols] */
int wft_colsum_bf16(const wft_bf16* x, int64_t rows, int64_t cols, int64_t ld,
                    float* out, int accumulate, void* stream);
/* The same sums for LARGE inputs (rows >= 8 192; round 6, was 65 536) through a caller workspace of wft_colsum_workspace_bytes(rows, cols): the
 * rows are cut into 64 chunks summed by (column group, chunk) workgroups and folded in chunk order (fixed order, HBM rate;
 * the one-pass kernel above occupies cols / 32 workgroups only).  Falls back to wft_colsum_bf16 when the workspace is
 * missing or the input small.  Replaces: the bias gradient torch.autograd forms for nn.Conv1d in
 * whisper.model.AudioEncoder (src/whisper_finetune/model/model_utils.py:271-281 via loss.backward(), :63-72).        */
int64_t wft_colsum_workspace_bytes(int64_t rows, int64_t cols);
int wft_colsum_bf16_ws(const wft_bf16* x, int64_t rows, int64_t cols, int64_t ld, float* out, int accumulate,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* -------------------------------------------------------------- LayerNorm */
/* whisper.model.LayerNorm.forward = F.layer_norm(x.float()).type(x.dtype)
 * (call sites model/model_utils.py:287,324 and every
 * ResidualAttentionBlock).  x,y bf16 [rows, cols]; gamma,beta f32[cols];
 * mean,rstd f32[rows] are saved for the backward.  cols % 8 == 0, <= 2048.
 * Optional deep-SpecAugment masks (model/model_utils.py:409-417): if
 * rows_per_batch > 0, rows t in [t0,t1) of every batch item and columns in
 * [c0,c1) are zero-filled in y (time mask then "frequency"=channel mask).   */
int wft_layernorm_fwd(const wft_bf16* x, const float* gamma, const float* beta,
                      wft_bf16* y, float* mean, float* rstd,
                      int64_t rows, int cols, float eps,
                      int rows_per_batch, int t0, int t1, int c0, int c1,
                      void* stream);
/* dx = LN'(dy) (+ dres if dres != NULL); dgamma/dbeta are WRITTEN (=; no zero-fill needed) or both NULL (frozen LayerNorm
 * parameters: with dx_colsum NULL as well — a LoRA run — the partial sums and the two reduce launches are skipped and
 * `partial` may be NULL).
 * partial: f32 workspace of wft_layernorm_bwd_workspace(rows, cols) bytes.
 * The same mask arguments zero the masked positions of dy first.
 * dx_colsum (f32 [cols] or NULL): column sums of the bf16 dx just written — dx is the gradient of the residual
 * stream, i.e. dy of the Linear (out / mlp.2) whose output was added to it, so this IS that Linear's bias
 * gradient and saves a separate pass over [rows, cols].                                          */
int64_t wft_layernorm_bwd_workspace(int64_t rows, int cols);
int wft_layernorm_bwd(const wft_bf16* dy, const wft_bf16* x, const float* gamma,
                      const float* mean, const float* rstd, const wft_bf16* dres,
                      wft_bf16* dx, float* dgamma, float* dbeta, float* dx_colsum, void* partial,
                      int64_t rows, int cols,
                      int rows_per_batch, int t0, int t1, int c0, int c1,
                      void* stream);

/* ------------------------------------------------------------------- GEMM */
/* Epilogue selector for wft_gemm_nt_bf16. */
enum {
  WFT_EPI_NONE = 0,
  WFT_EPI_GELU = 1,   /* C = gelu_erf(acc + bias); if aux != NULL also store pre-activation there */
  WFT_EPI_DGELU = 2,  /* C = acc * gelu'(aux)  (backward through GELU; aux = saved pre-activation) */
  /* The training pair of the MLP (mlp.0 -> GELU -> mlp.2): the forward GEMM stores gelu'(pre) instead of pre — same
   * bytes — and the backward-data GEMM multiplies by it, so the derivative's exp/rcp/polynomial (about 30 % of a
   * K = 1280 tile's time when done in the backward epilogue) is computed once, beside gelu() itself.  bf16 C only. */
  WFT_EPI_GELU_GRAD = 3, /* C = gelu_erf(acc + bias); aux (required) <- gelu'(acc + bias) */
  WFT_EPI_MUL_AUX = 4,   /* C = acc * aux */
  /* The same pair with gelu' stored in ONE BYTE per element (round 6): code = round(200 gelu') + 26, a 1/200 grid on which
   * gelu' = 0 and gelu' = 1 are exact, range [-0.13, 1.145] (gelu' lies in [-0.129, 1.129]), |error| <= 0.0025 — about the bf16
   * half-ulp near 1.  `aux` is then NOT an [M, N] matrix but an opaque buffer of wft_gemm_nt_aux8_bytes(args) bytes in the
   * FRAGMENT order of gemm_nt4w_kernel (16 KiB per 256x256 tile and wave; lda / strides of aux are ignored): the writer (mlp.0's
   * forward GEMM) and the reader (mlp.2's backward-data GEMM) have the same M and N, hence the same tiles.  Only the
   * one-wave-per-SIMD kernel carries them: ask wft_gemm_nt_aux8_bytes (0 = not served: use the bf16 pair).  batch 1, alpha 1.
   * Halves the bytes of the pair's second tensor on the store / load path that prices these epilogues, and saves one byte per
   * MLP activation element of saved-for-backward memory (21 GiB at 87 clips of large-v3).  Reference: `F.gelu` in
   * whisper.model.ResidualAttentionBlock.mlp (keys mlp.0 / mlp.2, scripts/convert_openai_to_hf.py:91-92).                     */
  WFT_EPI_GELU_GRAD8 = 5, /* C = gelu_erf(acc + bias); aux <- codes of gelu'(acc + bias) */
  WFT_EPI_MUL_AUX8 = 6    /* C = acc * decode(aux) */
};
/* C[b][m, n] = alpha * sum_k A[b][m, k] * B[b][n, k]  (+ bias[n]) (epilogue)
 *              (+ residual[b][m, n]);  bf16 inputs, fp32 MFMA accumulation.
 * Replaces whisper.model.Linear / Conv1d / the tied logits matmul
 * (model/model_utils.py:276-277,325).
 *  A: bf16, row m at A + b*strideA + m*lda, K contiguous (lda may be < K:
 *     overlapping rows = im2col-free conv1d windows);
 *  B: bf16 [N, K] K contiguous (a Linear weight as stored, or a transposed
 *     shadow for backward-data);  C: bf16 or f32 (c_is_f32), ldc;
 *  accumulate: C += result (only with c_is_f32);
 *  valid_rows_period/valid_rows: if period > 0, rows with (m % period) >=
 *     valid_rows are written as zero (keeps conv pad rows zero);
 *  K % 64 == 0, N % 128 == 0, M >= 1; all base pointers 16-byte aligned,
 *  lda/ldb % 8 == 0.                                                         */
typedef struct {
  const wft_bf16* A; int64_t lda; int64_t strideA;
  const wft_bf16* B; int64_t ldb; int64_t strideB;
  void* C; int64_t ldc; int64_t strideC; int c_is_f32; int accumulate;
  const float* bias;                    /* f32 [N] or NULL                    */
  const wft_bf16* residual; int64_t ldr; int64_t strideR;  /* or NULL         */
  wft_bf16* aux; int64_t ldaux; int64_t strideAux;         /* see epilogues    */
  int epilogue; float alpha;
  int64_t M; int64_t N; int64_t K; int batch;
  int valid_rows_period; int valid_rows;
  int residual_first;  /* != 0: add the residual BEFORE the epilogue op (gelu / dgelu) instead of after */
  /* wft_gemm_tn_bf16 only: optional caller-owned scratch for split-K partial tiles.  With at least
   * wft_gemm_tn_workspace_bytes(args) bytes the partials are written with plain stores and summed in a
   * fixed order by a second kernel (bitwise reproducible); with NULL / too small they are added to C
   * with fp32 atomics (order-dependent in the last bits).                                            */
  void* workspace; int64_t workspace_bytes;
  /* wft_gemm_nt_bf16 only: scale of the residual term, C = alpha*acc (+bias)(epilogue) + beta*residual.
   * 0 is read as 1 so that zero-initialised structs keep the plain residual add.  (Newton-Schulz steps of the
   * Muon optimizer: B = b*A + c*A@A, X' = a*X + B@X — muon.py zeropower_via_newtonschulz5.)            */
  float beta;
  /* wft_gemm_tn_bf16 only (P = 128, f32 C): if 0 < p_valid <= 64, columns >= p_valid of A are known to be zero (a rank-r LoRA
   * operand in its 128-wide padded buffer).  Selects the load-stream kernel for rank-r operands: only the first
   * 16*ceil(p_valid/16) columns of A are read, the split-K workspace holds only those rows (wft_gemm_tn_workspace_bytes
   * accounts for it), rows >= 16*ceil(p_valid/16) of C are written as zero (left alone when accumulating).  Results are
   * bit-identical to p_valid = 0.
   * wft_gemm_nt_bf16 (N = 128, bf16 C, batch 1, no bias / residual / epilogue): rows >= p_valid of B are zero padding (the
   * other two adapter products, u = x (sA*mask)^T and du = dy (sB)): only rows < 16*ceil(p_valid/16) of B are read and ONLY
   * those columns of C are written — the rest of the 128-wide buffer is left as it was (its consumer is the call above with the
   * same p_valid); bit-identical to p_valid = 0 in the written columns. */
  int p_valid;
  /* wft_gemm_nt_bf16 only: if != NULL (bf16 C, batch == 1), colsum[n] = sum over rows of the C just written — the bias
   * gradient of the Linear that consumes C as its dy (C = d(pre-activation) from the DGELU epilogue).  With `workspace`
   * of wft_gemm_nt_colsum_workspace_bytes(args) bytes the sums are formed in the epilogue of the 256x256 kernel (no
   * second pass over C); otherwise the library runs wft_colsum_bf16 over C after the GEMM.                      */
  float* colsum;
  /* wft_gemm_tn_bf16 with p_valid > 0 only (the two LoRA adapter gradients; both need the workspace, whose size
   * wft_gemm_tn_workspace_bytes reports with these fields set).  Applied by the kernel that sums the split-K partials, so the
   * adapter gradients leave the library in the layout autograd hands to the optimizer — no element-wise pass after the GEMM:
   *  tn_col_scale: f32 [S][Q] or NULL; C[p][q] *= tn_col_scale[p / tn_scale_rows][q] (tn_scale_rows = 0: S = 1, one row for
   *    all p).  dA = (du^T x) * mask, the dropout mask of each adapter of the group (model/lora.py via minLoRA's
   *    dropout-on-A).  In this form C is a [p_valid][ldc] matrix: rows >= p_valid are NOT written (the
   *    adapter gradients keep no 128-row buffer alive);
   *  tn_block_n > 0: C is NOT a [128][ldc] matrix but Q / tn_block_n contiguous blocks [tn_block_n][tn_block_r], block b
   *    holding the TRANSPOSE of rows b*tn_block_r .. +tn_block_r, columns b*tn_block_n .. +tn_block_n of the product (the
   *    off-diagonal blocks are not stored): dB of adapter b of a group of equally shaped Linears, as [out, r] row-major. */
  const float* tn_col_scale; int tn_scale_rows;
  int tn_block_n; int tn_block_r;
  /* wft_gemm_tn_bf16 on its 256x256 paths (ask wft_gemm_tn_segments_ok): tn_seg_count in 1..4 row ranges of the product go to
   * SEPARATE contiguous f32 [rows_i][Q] destinations instead of C — tn_seg_ptr[i] receives rows [tn_seg_end[i-1], tn_seg_end[i])
   * (tn_seg_end[count-1] = P) — with `accumulate` read-modify-write like C.  One weight-gradient GEMM of a fused Linear group
   * (q, k, v share one [3d, d] product) then writes each parameter's gradient where it lives: the `gradient_as_bucket_view` slices
   * of torch DDP's buckets (reference: scripts/finetune.py:698-705), no copy between the last dW GEMM and the bucket's all-reduce.
   * Always through the workspace (wft_gemm_tn_workspace_bytes with these fields set), summed by the reduce kernel in the same
   * fixed order as the unsegmented call: bit-identical values.  C must still be a valid pointer (it is not written).           */
  int tn_seg_count; int tn_seg_end[4]; float* tn_seg_ptr[4];
  /* Per-call launch state (zero = defaults; nothing is kept in the library):
   *  launch_mode: 0 = persistent grids for the 256x256 NT kernels (one workgroup per CU walks the tiles), 1 = one workgroup per tile
   *    (the backward pass beside RCCL's collective kernels: runtime.exchange_launch_mode; scripts/finetune.py:694-710).
   *  variant: != 0 forces the 8-wave ping-pong kernels (gemm_nt256_kernel / gemm_tn256_kernel) where the one-wave-per-SIMD kernels
   *    would apply (bit-identity tests, A/B runs; NT: same k order, bit-identical C).                                            */
  int launch_mode; int variant;
} wft_gemm_args;
int wft_gemm_nt_bf16(const wft_gemm_args* args, void* stream);
/* Which kernel wft_gemm_nt_bf16 dispatches these arguments to: 4 (gemm_nt4w_kernel: 256x256 tiles, four waves with 128x128
 * accumulators each), 256 (gemm_nt256_kernel, the 8-wave ping-pong 256x256 kernel) or 128 (gemm_nt_kernel).  Pure host
 * function (used by bench.py to attribute HIP-event timings to the kernel names rocprofv3 reports).        */
int wft_gemm_nt_variant(const wft_gemm_args* args);
/* 1 if wft_gemm_tn_bf16 takes these arguments' tn_seg_* fields (a 256x256 weight-gradient path, f32 C, no rank-r forms, a
 * well-formed segment list), 0 if the caller has to use C and copy.  Pure host function.                                    */
int wft_gemm_tn_segments_ok(const wft_gemm_args* args);
int64_t wft_gemm_nt_colsum_workspace_bytes(const wft_gemm_args* args);
/* Split-K form of the 128-tile kernel (round 6): bytes of fp32 partial tiles [nsplit][M][N] if a PLAIN bf16 product (batch 1, no bias /
 * residual / epilogue / colsum) has so few output tiles and so deep a K that the library would split K over the idle CUs when given
 * `workspace` of at least this size (the tied-embedding backward-data product of a short decoder batch, dX[B*S, d] = dlogits[B*S, V] E[V, d],
 * whisper.model.TextDecoder.forward's logits matmul reached from model_utils.py:83-84: 1 024 x 512 x 51 968 runs on 32 of 256 CUs
 * unsplit).  0: the call is not split.  Partials are summed in split order: bitwise reproducible.  Pure host function.          */
int64_t wft_gemm_nt_splitk_workspace_bytes(const wft_gemm_args* args);
/* Size of the one-byte gelu' buffer of WFT_EPI_GELU_GRAD8 / WFT_EPI_MUL_AUX8 for these arguments, or 0 if the call would not be
 * served in that form (not a gemm_nt4w_kernel problem).  Pure host function.                                                */
int64_t wft_gemm_nt_aux8_bytes(const wft_gemm_args* args);
/* C[p, q] (+)= alpha * sum_r A[r, p] * B[r, q]   (weight gradients dW = dY^T X:
 * what autograd's mm-backward computes for whisper.model.Linear).
 *  A bf16 [R, P] (row r at A + r*lda, P contiguous), B bf16 [R, Q];
 *  C f32 or bf16 [P, Q].  P % 128 == 0, Q % 128 == 0, any R >= 1.
 *  Uses the same wft_gemm_args: M:=P, N:=Q, K:=R; bias/residual/aux ignored.
 *  batch > 1 sums over the batch as extra reduction (conv weight grads).     */
int wft_gemm_tn_bf16(const wft_gemm_args* args, void* stream);
/* Two independent rank-r products (p_valid > 0) in ONE launch of the load-stream kernels — the backward of an adapted Linear
 * group in the reference's parametrization (src/whisper_finetune/model/lora.py:30-71 via minLoRA; loss.backward() at
 * model/model_utils.py:63-72) needs {u = x (sA*m)^T, du = dy (sB)} and then {dA = du^T x, dB^T = u^T dy}: each pair is one call.
 * Same arguments, same results (bit for bit) as two calls of wft_gemm_nt_bf16 / wft_gemm_tn_bf16, which is also the fallback
 * for anything the load-stream kernels do not take.  The TN pair needs a separate workspace per product
 * (wft_gemm_tn_workspace_bytes each); its two split-K reduces (column scale / block layout included) are one launch too.   */
int wft_gemm_nt_rank_pair_bf16(const wft_gemm_args* args0, const wft_gemm_args* args1, void* stream);
int wft_gemm_tn_rank_pair_bf16(const wft_gemm_args* args0, const wft_gemm_args* args1, void* stream);
int64_t wft_gemm_tn_workspace_bytes(const wft_gemm_args* args);

/* -------------------------------------------------------------- Attention */
/* whisper.model.MultiHeadAttention.qkv_attention:
 * softmax(q k^T / sqrt(64) [+ causal mask]) v, head_dim fixed to 64.
 * q: bf16, element (b, t, h, d) at q + b*q_bs + t*ldq + h*64 + d (so a fused
 * [B*T, 3*d_model] QKV buffer is consumed in place); same for k, v, o, do,
 * dq, dk, dv.  lse f32 [B, H, Tq] = log-sum-exp of the scaled scores (saved
 * for backward).  causal != 0: key j visible to query i iff j <= i.          */
typedef struct {
  const wft_bf16* q; int64_t ldq; int64_t q_bs;
  const wft_bf16* k; int64_t ldk; int64_t k_bs;
  const wft_bf16* v; int64_t ldv; int64_t v_bs;
  wft_bf16* o; int64_t ldo; int64_t o_bs;
  float* lse;
  int B; int H; int Tq; int Tk; int causal; float scale;
  /* backward only */
  const wft_bf16* d_o; int64_t lddo; int64_t do_bs;
  float* delta;                 /* f32 workspace [2, B, H, Tq] (kernel-internal contents: -rowsum(dO*O), then -lse/scale) */
  wft_bf16* dq; int64_t lddq; int64_t dq_bs;
  wft_bf16* dk; int64_t lddk; int64_t dk_bs;
  wft_bf16* dv; int64_t lddv; int64_t dv_bs;
  /* optional (backward): bias gradients of the q and v projections formed in the kernels' epilogues instead of a
   * second pass over dq / dv.  dq_colsum / dv_colsum: f32 [H*64] outputs (sum over batch and time of the bf16 values
   * written); colsum_ws: f32 scratch of wft_attn_bwd_colsum_workspace_bytes(args) bytes.  All three or none.      */
  float* dq_colsum; float* dv_colsum; float* colsum_ws;
  /* Per-call launch state (zero-initialised structs get the defaults; nothing here is kept in the library):
   *  launch_mode: 0 = the kernel's default grid (the dK/dV kernel: persistent, one workgroup per CU), 1 = one work item per workgroup
   *    (the backward pass that runs beside RCCL's collective kernels in a multi-process job: runtime.exchange_launch_mode; the
   *    reference's DDP wrap, scripts/finetune.py:694-710).  Same results either way.
   *  variant: bit 0 forward on attn_fwd_kernel, bit 1 dQ on the 8-wave kernel, bit 2 dK/dV on the 8-wave kernel even where the
   *    pipelined / one-wave-per-SIMD kernels apply (bit-identity tests, A/B runs).
   *  q_prescaled != 0: q (forward and backward) already holds (x Wq^T + bq) * scale * log2(e) — the factor folded into the bf16
   *    forward shadow of the q projection (wft_cast_pad_transpose_f32_bf16 / wft_lora_merge `fwd_scale`), ONE bf16 rounding of the
   *    scaled value, what `q * scale` costs upstream (whisper.model.MultiHeadAttention.qkv_attention) — so no kernel multiplies the
   *    scores.  `scale` must still be the softmax scale: dq is returned w.r.t. the UNSCALED projection output (dq = scale dS K),
   *    dk = dS^T q_prescaled * ln 2, lse stays the natural-log lse of the scaled scores.                                          */
  int launch_mode; int variant; int q_prescaled;
} wft_attn_args;
int wft_attn_fwd_bf16(const wft_attn_args* args, void* stream);
int wft_attn_bwd_bf16(const wft_attn_args* args, void* stream);
int64_t wft_attn_bwd_colsum_workspace_bytes(const wft_attn_args* args);
/* Which kernel wft_attn_fwd_bf16 / wft_attn_bwd_bf16 dispatch these arguments to (pure host function, like wft_gemm_nt_variant):
 * which = 0 forward: 2 attn_fwd_pipe_kernel, 1 attn_fwd_kernel; which = 1 dQ: 4 attn_bwd_dq4w_kernel, 8 attn_bwd_dq_kernel;
 * which = 2 dK/dV: 4 attn_bwd_dkdv4w_kernel, 8 attn_bwd_dkdv_kernel.  Only the shape / stride / causal fields are read.            */
int wft_attn_variant(const wft_attn_args* args, int which);

/* -------------------------------------------------------------- Embedding */
/* TextDecoder: x = token_embedding(tokens) + positional_embedding[:S]
 * (model/model_utils.py:317-318).  tokens i64 [B*S]; emb f32 [V, d];
 * pos f32 [n_ctx, d]; out bf16 [B*S, d].                                     */
int wft_embed_fwd(const int64_t* tokens, const float* emb, const float* pos,
                  wft_bf16* out, int64_t B, int64_t S, int d, int64_t V, void* stream);
/* demb[tokens[i], :] += dout[i, :] (f32 atomics), dpos[s, :] += sum_b dout.   */
int wft_embed_bwd(const int64_t* tokens, const wft_bf16* dout, float* demb, float* dpos,
                  int64_t B, int64_t S, int d, int64_t V, void* stream);

/* ---------------------------------------------------------- Cross entropy */
/* F.cross_entropy(logits.transpose(1,2), y_out, label_smoothing=eps),
 * ignore_index = -100, mean over non-ignored targets
 * (model/model_utils.py:66).  logits bf16 [rows, ld] (only the first V
 * columns are read); targets i64 [rows].
 * fwd: row_loss f32[rows] (0 for ignored rows), row_lse f32[rows],
 *      stats f32[2] = {sum of row losses, number of valid rows} (zeroed by
 *      the call, then accumulated) — loss = stats[0] / stats[1].
 * bwd: dlogits (bf16, may alias logits) = (softmax - ((1-eps) onehot + eps/V))
 *      * valid * gscale[0] / stats[1]; columns [V, ld) are written as 0.
 *      gscale: device f32[1] (the upstream grad of the mean loss).
 * argmax (optional, may be NULL): i64 [rows] = argmax over the V columns,
 *      lowest index on ties (eval/evaluator.py:70-73 teacher-forced argmax).  */
int wft_ce_fwd(const wft_bf16* logits, int64_t ld, const int64_t* targets,
               int64_t rows, int64_t V, float label_smoothing,
               float* row_loss, float* row_lse, float* stats, int64_t* argmax,
               void* stream);
int wft_ce_bwd(const wft_bf16* logits, int64_t ld, const int64_t* targets,
               int64_t rows, int64_t V, float label_smoothing,
               const float* row_lse, const float* stats, const float* gscale,
               wft_bf16* dlogits, void* stream);

/* Teacher-forced evaluation reductions (eval/evaluator.py:70-73 argmax;
 * eval/metrics.py:106-137 log_softmax / softmax / CE / entropy / max-prob) in ONE
 * pass over the logits: out4[row] = {lse, max logit, E_p[logit], target logit (0 if
 * the target is -100)}, argmax[row] = lowest index of the maximum.  targets may be
 * NULL.  nll = lse - x_t; log p(pred) = max - lse; confidence = exp(max - lse);
 * entropy = lse - E_p[logit].                                                  */
int wft_token_stats(const wft_bf16* logits, int64_t ld, const int64_t* targets,
                    int64_t rows, int64_t V, float* out4, int64_t* argmax, void* stream);

/* ------------------------------------------------- Log-mel + SpecAugment */
/* whisper.audio.log_mel_spectrogram (data/data_loader.py:278; SURVEY.md App.
 * A.2): reflect-pad 200, Hann-400 STFT hop 160, |.|^2, mel filterbank,
 * log10(clamp 1e-10), floor at clip max - 8, (x+4)/4.
 * audio f32 [B, n_samples] (n_samples = 160 * n_frames); filters f32
 * [n_mels, 201]; out f32 [B, n_mels, n_frames]; clipmax f32 [B] workspace.   */
int wft_logmel(const float* audio, const float* filters, float* out, float* clipmax,
               int B, int n_samples, int n_mels, int n_frames, void* stream);
/* SpecAugment on a batch of log-mels, all parameters already drawn on the
 * host in the reference's order (data/data_loader.py:284-290,
 * data/utils.py:41-143,146-190): per clip 8 ints
 *   {apply_warp, warp_p, warp_d, t0, t1, f0, f1, unused} and 2 ints
 *   {extreme_low_len, extreme_high_len}.
 * in/out f32 [B, n_mels, T]; must not alias.                                  */
int wft_specaug(const float* in, float* out, const int32_t* params, const int32_t* extremes,
                int B, int n_mels, int T, void* stream);
/* mel f32 [B, n_mels, T] -> time-major, channel-padded, row-padded bf16
 * [B, T+2, c_pad] with zero rows 0 and T+1 (the conv1d k=3 p=1 halo) — the
 * layout the conv-as-GEMM stem consumes (model/model_utils.py:276).          */
int wft_mel_to_tmajor_bf16(const float* mel, wft_bf16* out, int B, int n_mels, int T,
                           int c_pad, void* stream);
/* d(mel) is never needed (inputs carry no grad). */

/* ---------------------------------------------------------------- AdamW */
/* torch.optim.AdamW single-tensor step over a flat f32 range
 * (model/optimizer.py:240-262 → optimizer.step() in model_utils.py:122).
 * g is multiplied by gscale[0] (device scalar: the clip_grad_norm_ coefficient,
 * model_utils.py:107) before use.  If p_bf16 != NULL the bf16 shadow is
 * refreshed in the same pass.                                                */
int wft_adamw_step(float* p, const float* g, float* m, float* v, wft_bf16* p_bf16,
                   int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bias_corr1, float bias_corr2,
                   const float* gscale, void* stream);
/* out[0] += sum(g^2) over n elements (for the global grad norm).             */
int wft_sumsq_f32(const float* g, int64_t n, float* out, void* stream);

/* ------------------------------------------------ multi-tensor optimizer step */
/* One launch for a whole parameter list (model/optimizer.py:240-262, optimizer.step() at
 * model_utils.py:122, torch.nn.utils.clip_grad_norm_ at model_utils.py:107).
 * The tensor list is a caller-owned table in DEVICE memory:
 *   tab[row * n + t]  int64  address of tensor t's row-th array (row meaning per function)
 *   numel[t]          int64  elements of tensor t
 *   chunk_start[t]    int32  index of tensor t's first WFT_MT_CHUNK-element chunk; chunk_start[n] =
 *                            total_chunks = sum_t ceil(numel[t] / WFT_MT_CHUNK)                        */
#define WFT_MT_CHUNK 65536
/* out[0] = sum over all tensors of g^2 (tab row 0 = g, f32; a zero address = no gradient this step, contributes
 * nothing).  partial: f32 [total_chunks] scratch.  Two fixed-order stages: bitwise reproducible.   */
int wft_mt_sumsq_f32(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n,
                     int total_chunks, float* partial, float* out, void* stream);
/* torch.optim.AdamW step for every tensor (tab rows: 0 p, 1 g, 2 exp_avg, 3 exp_avg_sq; all f32).
 * If sumsq != NULL, g is first scaled by min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)): clip_grad_norm_
 * folded into the same pass (28 B/parameter of HBM traffic in total).                           */
int wft_mt_adamw(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n,
                 int total_chunks, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bias_corr1, float bias_corr2, const float* sumsq,
                 float max_norm, void* stream);

/* 8-bit AdamW: `bnb.optim.AdamW8bit` / `Adam8bit` of the reference's `optimizer.8bit: True` (model/optimizer.py:241-256; the
 * third-party package bitsandbytes is neither vendored nor installed: published algorithm, Dettmers et al. 2022 — PARITY
 * UNPINNED, see oracle/adam8bit_oracle.py).  The two moments are stored as one byte per element in blocks of
 * WFT_Q8_BLOCK = 2 048 elements with an f32 absmax per block; byte c decodes to qmap[c] * absmax.
 * tab rows: 0 p (f32), 1 g (f32), 2 state1 (u8 codes of m), 3 state2 (u8 codes of v), 4 absmax1 (f32 [ceil(numel / 2048)]),
 * 5 absmax2.  qmap1 / qmap2: the two ascending 256-entry f32 maps (signed / unsigned dynamic quantisation) in device memory.
 * Per block: dequantise, m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2 (g scaled by the clip coefficient as in wft_mt_adamw),
 * p += -lr * sqrt(bias_corr2) / bias_corr1 * m / (sqrt(v) + sqrt(bias_corr2) * eps), p *= 1 - lr * weight_decay, then the new
 * absmax of the block and the nearest map entry of m / absmax1, v / absmax2.  16 B / parameter of HBM traffic (28 with f32
 * state).  Chunks (WFT_MT_CHUNK elements) hold whole blocks, so blocks never span tensors.                               */
#define WFT_Q8_BLOCK 2048
int wft_mt_adamw8(const void* tab, const int64_t* numel, const int32_t* chunk_start, int n,
                  int total_chunks, const float* qmap1, const float* qmap2, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* sumsq,
                  float max_norm, void* stream);

/* --------------------------------------------------------------------- Muon */
/* The reference builds `muon.MuonWithAuxAdam` (third-party package `muon`, git HEAD, not vendored:
 * pyproject.toml:29, model/optimizer.py:171-237).  Its update for a 2-D parameter is
 *   buf = lerp(buf, g, 1-beta);  u = lerp(g, buf, beta) (nesterov);
 *   X = bf16(u) (transposed if rows > cols);  X /= ||X||_F + 1e-7;
 *   5 x { A = X X^T;  B = b A + c A A;  X = a X + B X }   (a, b, c) = (3.4445, -4.7750, 2.0315);
 *   p = p (1 - lr wd) - lr sqrt(max(1, rows/cols)) X.
 * The three products per iteration are wft_gemm_nt_bf16 launches (beta = residual scale) batched over all
 * parameters of one shape; the functions below are the element-wise / layout steps around them.
 * Tables as above with n = n_mats same-shape [rows, cols] matrices (tab rows: 0 p, 1 g, 2 momentum).   */
/* Step 1: momentum + nesterov; g is overwritten with u (as grad.lerp_ does); U [n][rows][cols] = bf16(u);
 * partial [n][ceil(rows*cols / WFT_MT_CHUNK)] = per-chunk sum of bf16(u)^2.  sumsq/max_norm as above.
 * A zero entry in table row 1 is a parameter without a gradient this step: treated as an all-zero gradient (the package
 * materialises one: muon.py "force synchronization"), nothing is written back for it.   */
int wft_muon_momentum_mt(const void* tab, int n_mats, int64_t numel, float beta, int nesterov,
                         wft_bf16* U, float* partial, const float* sumsq, float max_norm, void* stream);
/* Step 2: normalise (norm rounded to bf16 like torch's bf16 .norm()) and lay out for the GEMMs:
 * X [n][rows_pad][cols_pad] with rows <= cols (U^T when the parameter is tall), Xt its transpose
 * [n][cols_pad][rows_pad]; pads are written as zeros.                                            */
int wft_muon_prepare(const wft_bf16* U, int rows, int cols, const float* partial, int chunks,
                     wft_bf16* X, wft_bf16* Xt, int rows_pad, int cols_pad, int n_mats, void* stream);
/* dst[b][c][r] = src[b][r][c] (keeps X and X^T in step between Newton-Schulz iterations).          */
int wft_transpose_bf16(const wft_bf16* src, int rows, int cols, wft_bf16* dst, int batch, void* stream);
/* Step 4: p = p (1 - lr wd) - lr * scale * O[t][r][c]; O bf16, row stride ldo, matrix stride stride_o. */
int wft_muon_apply_mt(const void* tab, int n_mats, int rows, int cols, const wft_bf16* O, int64_t ldo,
                      int64_t stride_o, float lr, float weight_decay, float scale, void* stream);

/* ------------------------------------------------------------ fp32 compute mode */
/* The reference runs TRUE fp32 when AMP is off (`training.mixed_precision_training: False`: autocast disabled at
 * model/model_utils.py:64, eval/evaluator.py:69), and the north star asks for parity within 1e-3 relative in that mode.
 * These entry points are that mode (csrc/f32.hip): fp32 tensors, v_mfma_f32_32x32x2_f32 products, fp32 everywhere else,
 * fixed summation orders.  It is the parity mode of the small configurations, not a throughput path.
 *
 * C[b][m, n] = alpha * sum_k A(b; m, k) * B(b; k, n) (+ beta * C[b][m, n]) (+ bias[n]);
 *   A(b; m, k) = A[b * a_bs + m * a_rs + k * a_cs],  B(b; k, n) = B[b * b_bs + k * b_rs + n * b_cs]  (element strides:
 *   NT / TN / NN products and the conv stem's overlapping-window rows — lda < K — are all stride choices);
 *   C row-major with row stride ldc.  Replaces F.linear / F.conv1d / the attention matmuls / the tied logits matmul of
 *   whisper.model in fp32 (model/model_utils.py:276-281,316-325).                                   */
typedef struct {
  const float* A; int64_t a_rs; int64_t a_cs; int64_t a_bs;
  const float* B; int64_t b_rs; int64_t b_cs; int64_t b_bs;
  float* C; int64_t ldc; int64_t c_bs;
  const float* bias;          /* f32 [N] or NULL */
  int64_t M; int64_t N; int64_t K; int batch;
  float alpha; float beta;
} wft_gemm_f32_args;
int wft_gemm_f32(const wft_gemm_f32_args* args, void* stream);
/* In place: row r of s [nrows, cols] (row stride ld) <- softmax(scale * s[r, :]); with causal != 0 the row belongs to
 * query r % rows_per_mat and columns beyond it are masked (the decoder's -inf upper-triangular mask buffer).
 * qkv_attention's softmax(qk.float()).                                                             */
int wft_softmax_fwd_f32(float* s, int64_t nrows, int64_t cols, int64_t ld, float scale, int causal,
                        int64_t rows_per_mat, void* stream);
/* dp <- scale * p * (dp - sum_c p[r, c] * dp[r, c])  (gradient w.r.t. the un-scaled scores). */
int wft_softmax_bwd_f32(const float* p, float* dp, int64_t nrows, int64_t cols, int64_t ld, float scale, void* stream);
/* whisper.model.LayerNorm in fp32 (+ the deep-SpecAugment mask, model/model_utils.py:409-417):
 * mask = int32 {rows_per_batch, t0, t1, c0, c1} in HOST memory or NULL.                                                 */
int wft_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int64_t rows, int cols, float eps, const int32_t* mask, void* stream);
int wft_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          float* dx, float* dgamma, float* dbeta, int64_t rows, int cols, const int32_t* mask, void* stream);
int wft_gelu_fwd_f32(const float* x, float* y, int64_t n, void* stream);                 /* exact-erf GELU             */
int wft_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int wft_axpby_f32(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream); /* y may be NULL */
int wft_colsum_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, float* out, void* stream);     /* bias grads   */
/* out[b, s, :] = emb[tokens[b, s], :] + pos[s, :]  (model/model_utils.py:316-318); the backward needs demb zero-filled. */
int wft_embed_fwd_f32(const int64_t* tokens, const float* emb, const float* pos, float* out, int64_t B, int64_t S, int d,
                      void* stream);
int wft_embed_bwd_f32(const int64_t* tokens, const float* dout, float* demb, float* dpos, int64_t B, int64_t S, int d,
                      void* stream);
/* F.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing) on fp32 logits (model/model_utils.py:66): row losses,
 * row logsumexp, stats = {sum of row losses, number of non-ignored rows}; the backward overwrites the logits with
 * gscale / n_valid * (softmax - (1 - eps) onehot - eps / V).                                                            */
int wft_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V, float label_smoothing,
                   float* row_loss, float* row_lse, float* stats, void* stream);
int wft_ce_bwd_f32(float* logits, int64_t ld, const int64_t* targets, int64_t rows, int64_t V, float label_smoothing,
                   const float* row_lse, const float* stats, const float* gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WFT_H */
