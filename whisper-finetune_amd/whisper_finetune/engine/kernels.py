"""Tensor-level wrappers over the C ABI (no autograd here; see ops.py).

Every function takes torch CUDA tensors, checks dtype/contiguity, and enqueues the HIP
kernel on the current stream.  Shapes follow include/wft.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L

BF16 = torch.bfloat16
F32 = torch.float32

# bench.py sets this to a list to time every gemm_nt launch with HIP events recorded on the
# launch stream: entries are (start_event, end_event, algorithmic_flops, kernel variant 128|256, algorithmic_bytes).
PROFILE_NT = None


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise L.WftError(f"{name}: libwft kernels need CUDA/HIP tensors (got {t.device}); there is no CPU path")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------- casts
def cast_bf16(src: torch.Tensor) -> torch.Tensor:
    _chk(src, F32, "src")
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    L.check(L.load().wft_cast_f32_bf16(_p(src), _p(dst), src.numel(), L.stream_ptr()), "wft_cast_f32_bf16")
    return dst


def cast_f32(src: torch.Tensor) -> torch.Tensor:
    _chk(src, BF16, "src")
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=F32, device=src.device)
    L.check(L.load().wft_cast_bf16_f32(_p(src), _p(dst), src.numel(), L.stream_ptr()), "wft_cast_bf16_f32")
    return dst


def weight_shadow(w: torch.Tensor, rows_pad: int, cols_pad: int, want_t: bool, out=None, out_t=None, fwd_scale: float = 1.0):
    """f32 [rows, cols] -> bf16 [rows_pad, cols_pad] (+ transposed [cols_pad, rows_pad]).  fwd_scale: the straight image is
    bf16(fwd_scale * w) (one rounding), the transposed one stays bf16(w) (wft.h; ops.QK_PRESCALE)."""
    _chk(w, F32, "w")
    w2 = w.reshape(w.shape[0], -1).contiguous()
    rows, cols = w2.shape
    dst = out if out is not None else torch.empty((rows_pad, cols_pad), dtype=BF16, device=w.device)
    dst_t = None
    if want_t:
        dst_t = out_t if out_t is not None else torch.empty((cols_pad, rows_pad), dtype=BF16, device=w.device)
    assert dst.stride(1) == 1 and (dst_t is None or dst_t.stride(1) == 1)
    L.check(
        L.load().wft_cast_pad_transpose_f32_bf16(_p(w2), rows, cols, _p(dst), _p(dst_t), rows_pad, cols_pad,
                                                 dst.stride(0), 0 if dst_t is None else dst_t.stride(0), float(fwd_scale), L.stream_ptr()),
        "wft_cast_pad_transpose_f32_bf16",
    )
    return dst, dst_t


def lora_merge(w, B, A, mask, scaling: float, rows_pad=None, cols_pad=None, out=None, out_t=None, out_f32=None, fwd_scale: float = 1.0):
    """W + scaling * B @ (A * mask) -> bf16 shadow `out` [rows_pad, cols_pad] (+ `out_t`) and/or f32 `out_f32` [rows, cols]."""
    for n, t in (("w", w), ("B", B), ("A", A)):
        _chk(t, F32, n)
    w2 = w.reshape(w.shape[0], -1).contiguous()
    rows, cols = w2.shape
    B, A = B.contiguous(), A.contiguous()
    r = A.shape[0]
    assert B.shape == (rows, r) and A.shape == (r, cols)
    if mask is not None:
        _chk(mask, F32, "mask")
        mask = mask.reshape(-1).contiguous()
        assert mask.numel() == cols
    if out is not None:
        rows_pad = out.shape[0] if rows_pad is None else rows_pad
        cols_pad = out.shape[1] if cols_pad is None else cols_pad
        assert out.stride(1) == 1 and (out_t is None or out_t.stride(1) == 1)
    L.check(
        L.load().wft_lora_merge(_p(w2), rows, cols, _p(B), _p(A), _p(mask), r, float(scaling), _p(out), _p(out_t),
                                rows_pad or rows, cols_pad or cols, 0 if out is None else out.stride(0),
                                0 if out_t is None else out_t.stride(0), _p(out_f32), float(fwd_scale), L.stream_ptr()),
        "wft_lora_merge",
    )
    return out, out_t, out_f32


def lora_refresh_mt(table: torch.Tensor, tile_start: torch.Tensor, n: int, total_tiles: int) -> None:
    """wft_lora_refresh_mt: lora_merge (bf16 shadow + transposed shadow) and lora_pack of every adapter in the device table
    (int64 [n, 20], see wft.h) in one launch."""
    L.check(L.load().wft_lora_refresh_mt(_p(table), _p(tile_start), int(n), int(total_tiles), L.stream_ptr()), "wft_lora_refresh_mt")


def mt_copy_f32(table: torch.Tensor) -> None:
    """wft_mt_copy_f32: every (source, destination, count) row of the device table int64 [n, 3] in one launch."""
    L.check(L.load().wft_mt_copy_f32(_p(table), int(table.shape[0]), L.stream_ptr()), "wft_mt_copy_f32")


def lora_pack(A, mask, B, scaling: float, Am, AmT, Bb, BbT, ro: int, no: int) -> None:
    """One adapter's blocks of the rank-r gradient-GEMM operands (see wft_lora_pack): A f32 [r, K], mask f32 [1, K] or None,
    B f32 [n, r] -> Am [Rpad, K] / AmT, Bb [Npad, Rpad] / BbT (bf16, zero-initialised by the caller)."""
    _chk(A, F32, "A"); _chk(B, F32, "B")
    for t, nme in ((Am, "Am"), (AmT, "AmT"), (Bb, "Bb"), (BbT, "BbT")):
        _chk(t, BF16, nme)
    r, Kd = A.shape
    n = B.shape[0]
    assert A.is_contiguous() and B.is_contiguous() and B.shape[1] == r and Am.shape[1] == Kd
    rpad, npad = Am.shape[0], Bb.shape[0]
    m = None if mask is None else mask.contiguous()
    L.check(L.load().wft_lora_pack(_p(A), _p(m), _p(B), r, Kd, n, float(scaling), _p(Am), _p(AmT), _p(Bb), _p(BbT), rpad, npad, ro, no,
                                   L.stream_ptr()), "wft_lora_pack")


def add_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    a = a.contiguous(); b = b.contiguous()
    y = torch.empty_like(a)
    L.check(L.load().wft_add_bf16(_p(a), _p(b), _p(y), a.numel(), L.stream_ptr()), "wft_add_bf16")
    return y


def axpby_bf16(a: float, x: torch.Tensor, b: float = 0.0, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a*x + b*y (bf16, same shape; y optional)."""
    _chk(x, BF16, "x")
    x = x.contiguous()
    if y is not None:
        _chk(y, BF16, "y")
        y = y.contiguous()
    out = torch.empty_like(x)
    L.check(L.load().wft_axpby_bf16(float(a), _p(x), float(b), _p(y), _p(out), x.numel(), L.stream_ptr()), "wft_axpby_bf16")
    return out


def dgelu_mul(dy: torch.Tensor, pre: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(dy, BF16, "dy"); _chk(pre, BF16, "pre")
    assert dy.is_contiguous() and pre.is_contiguous()
    out = torch.empty_like(dy) if out is None else out
    L.check(L.load().wft_dgelu_mul_bf16(_p(dy), _p(pre), _p(out), dy.numel(), L.stream_ptr()), "wft_dgelu_mul_bf16")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """x bf16 [rows, cols] (row stride may exceed cols) -> f32 [cols]."""
    _chk(x, BF16, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=F32, device=x.device)
        accumulate = False
    lib = L.load()
    need = lib.wft_colsum_workspace_bytes(rows, cols)
    if need > 0:  # large inputs: chunked over the whole chip, folded in a fixed order
        ws = _tn_workspace(x.device, need, slot="colsum")
        L.check(lib.wft_colsum_bf16_ws(_p(x), rows, cols, x.stride(0), _p(out), int(accumulate), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                "wft_colsum_bf16_ws")
    else:
        L.check(lib.wft_colsum_bf16(_p(x), rows, cols, x.stride(0), _p(out), int(accumulate), L.stream_ptr()), "wft_colsum_bf16")
    return out


# --------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps=1e-5, mask=None):
    """x bf16 [..., cols]; mask = (rows_per_batch, t0, t1, c0, c1) or None."""
    _chk(x, BF16, "x"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta")
    x = x.contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    rpb, t0, t1, c0, c1 = mask if mask is not None else (0, 0, 0, 0, 0)
    L.check(
        L.load().wft_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols, eps,
                                   rpb, t0, t1, c0, c1, L.stream_ptr()),
        "wft_layernorm_fwd",
    )
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, mask=None, want_colsum=False, want_params=True):
    """returns dx bf16, dgamma f32, dbeta f32 (fresh tensors, each its own allocation so that autograd can keep them as
    the .grad without a copy; None, None with want_params=False: frozen LayerNorm parameters) [, colsum(dx) f32 if want_colsum]."""
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x")
    dy = dy.contiguous(); x = x.contiguous()
    if dres is not None:
        _chk(dres, BF16, "dres")
        dres = dres.contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols
    lib = L.load()
    need_ws = want_params or want_colsum
    ws = torch.empty(lib.wft_layernorm_bwd_workspace(rows, cols), dtype=torch.uint8, device=x.device) if need_ws else None
    dx = torch.empty_like(x)
    dgamma = torch.empty(cols, dtype=F32, device=x.device) if want_params else None
    dbeta = torch.empty(cols, dtype=F32, device=x.device) if want_params else None
    dxs = torch.empty(cols, dtype=F32, device=x.device) if want_colsum else None
    rpb, t0, t1, c0, c1 = mask if mask is not None else (0, 0, 0, 0, 0)
    L.check(
        lib.wft_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma), _p(dbeta),
                              _p(dxs), _p(ws), rows, cols, rpb, t0, t1, c0, c1, L.stream_ptr()),
        "wft_layernorm_bwd",
    )
    if want_colsum:
        return dx, dgamma, dbeta, dxs
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- GEMM
# Per-call launch state (wft.h: wft_gemm_args / wft_attn_args `launch_mode`, `variant`, `q_prescaled`).  libwft keeps NO mutable state;
# these Python-side hooks only choose what the wrappers below write into the argument structs.
#   VARIANT: test / A-B hook — force the 8-wave kernels ("nt", "tn": gemm_nt256 / gemm_tn256; "fwd", "dq", "dkdv": the 8-wave
#     attention kernels) where the one-wave-per-SIMD kernels would apply; `set_variant` returns the previous value.
#   LAUNCH_OVERRIDE[0]: bench tooling (bench.py ddp_mode_1gpu) — None: the caller's `launch` argument decides; 0 / 1: every call.
VARIANT = {"nt": 0, "tn": 0, "fwd": 0, "dq": 0, "dkdv": 0}
LAUNCH_OVERRIDE = [None]
LAUNCH_PERSISTENT, LAUNCH_PER_TILE = 0, 1


def set_variant(which: str, v: int) -> int:
    old = VARIANT[which]
    if v >= 0:
        VARIANT[which] = 1 if v else 0
    return old


def _launch(launch: int) -> int:
    o = LAUNCH_OVERRIDE[0]
    return int(launch) if o is None else int(o)


def _attn_variant_bits() -> int:
    return VARIANT["fwd"] | (VARIANT["dq"] << 1) | (VARIANT["dkdv"] << 2)


def gemm_nt(a, b, *, M=None, N=None, K=None, lda=None, ldb=None, out=None, out_f32=False, accumulate=False,
            bias=None, residual=None, aux=None, epilogue=L.EPI_NONE, alpha=1.0, batch=1,
            strideA=0, strideB=0, strideC=0, strideR=0, strideAux=0, ldc=None,
            valid_rows_period=0, valid_rows=0, residual_first=False, ldaux=None, ldr=None, beta=1.0, colsum=None, p_valid=0,
            launch=0, _args_only=False):
    """C[M,N] = alpha * A[M,K] @ B[N,K]^T (+bias) (epilogue) (+residual).

    a: bf16, row m at a.data_ptr() + m*lda; b: bf16 [N, K] (ldb).  Defaults take the shapes
    from 2-D contiguous tensors.  `aux`: GELU pre-activation out (EPI_GELU) / in (EPI_DGELU); gelu' out (EPI_GELU_GRAD) / in (EPI_MUL_AUX).
    p_valid (N = 128, plain bf16 product): rows >= p_valid of b are zero padding (a rank-r LoRA operand) — the load-stream
    kernel for rank-r operands, which writes ONLY columns < 16*ceil(p_valid/16) of the output (the rest stays uninitialised:
    the consumer is gemm_tn with the same p_valid).
    """
    _chk(a, BF16, "A"); _chk(b, BF16, "B")
    if M is None:
        M = a.shape[0]
    if K is None:
        K = a.shape[-1]
    if N is None:
        N = b.shape[0]
    if lda is None:
        lda = a.stride(-2) if a.dim() >= 2 else K
    if ldb is None:
        ldb = b.stride(-2)
    if out is None:
        out = torch.empty((batch * M, N) if batch > 1 else (M, N), dtype=F32 if out_f32 else BF16, device=a.device)
        if batch > 1 and strideC == 0:
            strideC = M * N
    if ldc is None:
        ldc = out.stride(-2)
    out_f32 = out.dtype == F32
    args = L.GemmArgs()
    args.A, args.lda, args.strideA = a.data_ptr(), lda, strideA
    args.B, args.ldb, args.strideB = b.data_ptr(), ldb, strideB
    args.C, args.ldc, args.strideC = out.data_ptr(), ldc, strideC
    args.c_is_f32, args.accumulate = int(out_f32), int(accumulate)
    args.bias = 0 if bias is None else bias.data_ptr()
    if residual is not None:
        _chk(residual, BF16, "residual")
        args.residual, args.ldr, args.strideR = residual.data_ptr(), (residual.stride(-2) if ldr is None else ldr), strideR
    if aux is not None and epilogue in (L.EPI_GELU_GRAD8, L.EPI_MUL_AUX8):
        # one-byte gelu' in gemm_nt4w_kernel's fragment order (wft.h): an opaque uint8 buffer of aux8_bytes(...) bytes
        _chk(aux, torch.uint8, "aux")
        args.aux, args.ldaux, args.strideAux = aux.data_ptr(), 0, 0
    elif aux is not None:
        _chk(aux, BF16, "aux")
        args.aux, args.ldaux, args.strideAux = aux.data_ptr(), (aux.stride(-2) if ldaux is None else ldaux), strideAux
    args.epilogue, args.alpha, args.beta = epilogue, alpha, beta
    args.M, args.N, args.K, args.batch = M, N, K, batch
    args.valid_rows_period, args.valid_rows = valid_rows_period, valid_rows
    args.residual_first = int(residual_first)
    args.p_valid = int(p_valid)
    args.launch_mode, args.variant = _launch(launch), VARIANT["nt"]
    if colsum is not None:  # f32 [N]: column sums of C (bias gradient of C's consumer), fused into the epilogue when possible
        _chk(colsum, F32, "colsum")
        args.colsum = colsum.data_ptr()
        need = L.load().wft_gemm_nt_colsum_workspace_bytes(C.byref(args))
        if need > 0:
            ws = _tn_workspace(a.device, need, slot="nt_colsum")
            args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    elif not out_f32 and batch == 1 and epilogue == L.EPI_NONE and K >= 4096 and N * M <= (1 << 21):
        # few output tiles over a deep K (the tied-embedding backward-data product of a short decoder batch): split-K partials
        need = L.load().wft_gemm_nt_splitk_workspace_bytes(C.byref(args))
        if need > 0:
            ws = _tn_workspace(a.device, need, slot="nt_splitk")
            args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    if _args_only:  # (paired launches: gemm_nt_rank_pair)
        return args, out
    if epilogue in (L.EPI_GELU_GRAD8, L.EPI_MUL_AUX8):
        need = L.load().wft_gemm_nt_aux8_bytes(C.byref(args))
        if need <= 0 or aux is None or aux.numel() < need:
            raise L.WftError(f"one-byte gelu' epilogue: the call is not served in that form or the buffer is too small (need {need} bytes)")
    if PROFILE_NT is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(L.load().wft_gemm_nt_bf16(C.byref(args), L.stream_ptr()), "wft_gemm_nt_bf16")
        ev1.record()
        nbytes = 2.0 * batch * (M * K + N * K + M * N * (2 if out_f32 else 1) + (M * N if residual is not None else 0) + (M * N if aux is not None else 0))
        PROFILE_NT.append((ev0, ev1, 2.0 * M * N * K * batch, L.load().wft_gemm_nt_variant(C.byref(args)), nbytes))
        return out
    L.check(L.load().wft_gemm_nt_bf16(C.byref(args), L.stream_ptr()), "wft_gemm_nt_bf16")
    return out


def gemm_nt_aux8_bytes(M: int, N: int, K: int, device, epilogue=None, colsum: bool = False) -> int:
    """Bytes of the one-byte gelu' buffer if an [M, K] x [N, K]^T product with the GELU_GRAD8 / MUL_AUX8 epilogue is served by
    gemm_nt4w_kernel (wft_gemm_nt_aux8_bytes), else 0.  Shapes only — the pointers are placeholders with the alignment real operands have."""
    args = L.GemmArgs()
    args.A = args.B = args.C = args.aux = 1 << 20
    args.lda, args.ldb, args.ldc = K, K, N
    args.M, args.N, args.K, args.batch, args.alpha, args.beta = M, N, K, 1, 1.0, 1.0
    args.epilogue = L.EPI_GELU_GRAD8 if epilogue is None else epilogue
    args.variant = VARIANT["nt"]
    if colsum:
        args.colsum = 1 << 20
    return int(L.load().wft_gemm_nt_aux8_bytes(C.byref(args)))


def gemm_tn(a, b, *, P=None, Q=None, R=None, lda=None, ldb=None, out=None, out_f32=True, accumulate=False,
            alpha=1.0, batch=1, strideA=0, strideB=0, p_valid=0, col_scale=None, scale_rows=0, block_n=0, block_r=0,
            seg_out=None, _args_only=False, _ws_slot="tn"):
    """C[P,Q] (+)= alpha * A[R,P]^T @ B[R,Q]  (weight gradients).  p_valid (P = 128 only): columns >= p_valid of A are zero
    padding (a rank-r LoRA operand): the reduction runs in the load-stream kernel for rank-r operands (gemm.hip
    gemm_tn_rank_kernel — 4-stage LDS-DMA ring, compact A, compact split-K workspace), bit-identical to the general path.
    With p_valid the kernel that sums the split-K partials can also finish the two LoRA adapter gradients (wft.h tn_col_scale /
    tn_block_n): col_scale f32 [S, Q] multiplies C[p, q] by col_scale[p // scale_rows, q] (dA = (du^T x) * mask);
    block_n / block_r return a flat f32 tensor of Q / block_n blocks [block_n, block_r], block b = the transpose of rows
    b*block_r.., columns b*block_n.. of the product (dB of adapter b as [out, r] row-major).
    seg_out: 1..4 contiguous f32 [rows_i, Q] tensors that receive consecutive row ranges of the product INSTEAD of one [P, Q]
    tensor (wft.h tn_seg_*: each parameter of a fused Linear group gets its gradient where it lives — a DDP bucket view);
    returns None if the library does not take the call in that form (the caller falls back to `out` + slicing), else seg_out."""
    _chk(a, BF16, "A"); _chk(b, BF16, "B")
    if R is None:
        R = a.shape[0]
    if P is None:
        P = a.shape[1]
    if Q is None:
        Q = b.shape[1]
    if lda is None:
        lda = a.stride(0)
    if ldb is None:
        ldb = b.stride(0)
    if seg_out is not None:
        out, ldc = seg_out[0], Q
    elif block_n:
        if out is None:
            out = torch.empty(Q * block_r, dtype=F32, device=a.device)
            accumulate = False
        ldc = Q
    else:
        if out is None:
            # col_scale form (dA of a group's adapters): the library writes p_valid rows only — a compact gradient tensor
            rows = int(p_valid) if (col_scale is not None and p_valid > 0) else P
            out = torch.empty((rows, Q), dtype=F32 if out_f32 else BF16, device=a.device)
            accumulate = False
        ldc = out.stride(0)
    args = L.GemmArgs()
    args.A, args.lda, args.strideA = a.data_ptr(), lda, strideA
    args.B, args.ldb, args.strideB = b.data_ptr(), ldb, strideB
    args.C, args.ldc, args.strideC = out.data_ptr(), ldc, 0
    args.c_is_f32, args.accumulate = int(out.dtype == F32), int(accumulate)
    args.epilogue, args.alpha = L.EPI_NONE, alpha
    args.M, args.N, args.K, args.batch = P, Q, R, batch
    args.p_valid = int(p_valid)
    if col_scale is not None:
        _chk(col_scale, F32, "col_scale")
        args.tn_col_scale, args.tn_scale_rows = col_scale.data_ptr(), int(scale_rows)
    args.tn_block_n, args.tn_block_r = int(block_n), int(block_r)
    args.variant = VARIANT["tn"]
    lib = L.load()
    if seg_out is not None:
        end = 0
        args.tn_seg_count = len(seg_out)
        for i, t in enumerate(seg_out):
            end += t.shape[0]
            args.tn_seg_end[i], args.tn_seg_ptr[i] = end, t.data_ptr()
        if not lib.wft_gemm_tn_segments_ok(C.byref(args)):
            return None
    need = lib.wft_gemm_tn_workspace_bytes(C.byref(args))
    if need > 0:
        ws = _tn_workspace(a.device, need, slot=_ws_slot)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    if _args_only:  # (paired launches: gemm_tn_rank_pair)
        return args, out
    L.check(lib.wft_gemm_tn_bf16(C.byref(args), L.stream_ptr()), "wft_gemm_tn_bf16")
    return out if seg_out is None else seg_out


def gemm_nt_rank_pair(a0, b0, a1, b1, p_valid: int):
    """(a0 @ b0^T, a1 @ b1^T) for two rank-r operands b0, b1 ([128, K] padded buffers, p_valid data rows) in ONE launch of the
    load-stream kernel (wft_gemm_nt_rank_pair_bf16): u = x (sA*m)^T and du = dy (sB) of an adapted Linear group's backward."""
    args0, out0 = gemm_nt(a0, b0, p_valid=p_valid, _args_only=True)
    args1, out1 = gemm_nt(a1, b1, p_valid=p_valid, _args_only=True)
    L.check(L.load().wft_gemm_nt_rank_pair_bf16(C.byref(args0), C.byref(args1), L.stream_ptr()), "wft_gemm_nt_rank_pair_bf16")
    return out0, out1


def gemm_tn_rank_pair(kw0: dict, kw1: dict):
    """Two gemm_tn(**kw) products with rank-r A operands (p_valid) in ONE launch plus ONE split-K reduce launch
    (wft_gemm_tn_rank_pair_bf16): dA = (du^T x) * mask and dB = u^T dy of an adapted Linear group.  Each product has its own
    workspace."""
    args0, out0 = gemm_tn(**kw0, _args_only=True, _ws_slot="tn")
    args1, out1 = gemm_tn(**kw1, _args_only=True, _ws_slot="tn_b")
    L.check(L.load().wft_gemm_tn_rank_pair_bf16(C.byref(args0), C.byref(args1), L.stream_ptr()), "wft_gemm_tn_rank_pair_bf16")
    return out0, out1


_TN_WS = {}


def _tn_workspace(device, nbytes: int, slot: str = "tn") -> torch.Tensor:
    """Per-device scratch for deterministic split-K weight-gradient GEMMs / fused column sums (grown on demand,
    reused: all launches are ordered on one stream)."""
    key = (device.type, device.index, slot)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _TN_WS[key] = ws
    return ws


# --------------------------------------------------------------------------- attention
def _attn_view(t: torch.Tensor):
    """t: [B, T, H*64-ish view] with last-dim stride 1 -> (ptr, ld, batch_stride)."""
    assert t.dim() == 3 and t.stride(2) == 1
    return t.data_ptr(), t.stride(1), t.stride(0)


def attn_fwd(q, k, v, n_head: int, causal: bool, scale: float, q_prescaled: bool = False):
    """q [B,Tq,H*64], k/v [B,Tk,H*64] bf16 views (any row stride) -> o [B,Tq,H*64], lse [B,H,Tq].
    q_prescaled: q already carries scale * log2(e) (ops.QK_PRESCALE folded into the q projection's forward shadow)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _chk(t, BF16, n)
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert D == n_head * 64, "head_dim must be 64"
    o = torch.empty((B, Tq, D), dtype=BF16, device=q.device)
    lse = torch.empty((B, n_head, Tq), dtype=F32, device=q.device)
    a = L.AttnArgs()
    a.q, a.ldq, a.q_bs = _attn_view(q)
    a.k, a.ldk, a.k_bs = _attn_view(k)
    a.v, a.ldv, a.v_bs = _attn_view(v)
    a.o, a.ldo, a.o_bs = _attn_view(o)
    a.lse = lse.data_ptr()
    a.B, a.H, a.Tq, a.Tk, a.causal, a.scale = B, n_head, Tq, Tk, int(causal), scale
    a.variant, a.q_prescaled = _attn_variant_bits(), int(bool(q_prescaled))
    L.check(L.load().wft_attn_fwd_bf16(C.byref(a), L.stream_ptr()), "wft_attn_fwd_bf16")
    return o, lse


def attn_bwd(q, k, v, o, lse, do, n_head: int, causal: bool, scale: float, dq=None, dk=None, dv=None, colsums=None,
             q_prescaled: bool = False, launch: int = 0):
    """colsums = (cs_q, cs_v): optional f32 [H*64] outputs, the column sums over (batch, time) of dq / dv (bias gradients).
    q_prescaled: as in attn_fwd; dq is the gradient w.r.t. the UNSCALED projection output either way."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    _chk(do, BF16, "do")
    if do.stride(2) != 1:
        do = do.contiguous()
    dq = torch.empty((B, Tq, D), dtype=BF16, device=q.device) if dq is None else dq
    dk = torch.empty((B, Tk, D), dtype=BF16, device=q.device) if dk is None else dk
    dv = torch.empty((B, Tk, D), dtype=BF16, device=q.device) if dv is None else dv
    delta = torch.empty((2, B, n_head, Tq), dtype=F32, device=q.device)  # kernel-internal workspace: [0] -rowsum(dO*O), [1] -lse/scale
    a = L.AttnArgs()
    a.q, a.ldq, a.q_bs = _attn_view(q)
    a.k, a.ldk, a.k_bs = _attn_view(k)
    a.v, a.ldv, a.v_bs = _attn_view(v)
    a.o, a.ldo, a.o_bs = _attn_view(o)
    a.lse = lse.data_ptr()
    a.B, a.H, a.Tq, a.Tk, a.causal, a.scale = B, n_head, Tq, Tk, int(causal), scale
    a.d_o, a.lddo, a.do_bs = _attn_view(do)
    a.delta = delta.data_ptr()
    a.dq, a.lddq, a.dq_bs = _attn_view(dq)
    a.dk, a.lddk, a.dk_bs = _attn_view(dk)
    a.dv, a.lddv, a.dv_bs = _attn_view(dv)
    a.variant, a.q_prescaled, a.launch_mode = _attn_variant_bits(), int(bool(q_prescaled)), _launch(launch)
    if colsums is not None:
        cs_q, cs_v = colsums
        _chk(cs_q, F32, "cs_q"); _chk(cs_v, F32, "cs_v")
        assert cs_q.numel() == D and cs_v.numel() == D and cs_q.is_contiguous() and cs_v.is_contiguous()
        ws = _tn_workspace(q.device, L.load().wft_attn_bwd_colsum_workspace_bytes(C.byref(a)), slot="attn_colsum")
        a.dq_colsum, a.dv_colsum, a.colsum_ws = cs_q.data_ptr(), cs_v.data_ptr(), ws.data_ptr()
    L.check(L.load().wft_attn_bwd_bf16(C.byref(a), L.stream_ptr()), "wft_attn_bwd_bf16")
    return dq, dk, dv


# --------------------------------------------------------------------------- embedding / CE
def embed_fwd(tokens, emb, pos):
    _chk(tokens, torch.int64, "tokens"); _chk(emb, F32, "emb"); _chk(pos, F32, "pos")
    tokens = tokens.contiguous()
    B, S = tokens.shape
    V, d = emb.shape
    out = torch.empty((B, S, d), dtype=BF16, device=emb.device)
    L.check(L.load().wft_embed_fwd(_p(tokens), _p(emb), _p(pos), _p(out), B, S, d, V, L.stream_ptr()), "wft_embed_fwd")
    return out


def embed_bwd(tokens, dout, demb, dpos):
    """accumulates into demb f32 [V,d] and dpos f32 [n_ctx,d]."""
    _chk(dout, BF16, "dout")
    tokens = tokens.contiguous(); dout = dout.contiguous()
    B, S = tokens.shape
    V, d = demb.shape
    L.check(L.load().wft_embed_bwd(_p(tokens), _p(dout), _p(demb), _p(dpos), B, S, d, V, L.stream_ptr()), "wft_embed_bwd")


def ce_fwd(logits, targets, V: int, label_smoothing: float, want_argmax: bool = False):
    """logits bf16 [rows, ld>=V]; targets i64 [rows] -> (row_loss, row_lse, stats[2], argmax|None)."""
    _chk(logits, BF16, "logits"); _chk(targets, torch.int64, "targets")
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, ld = logits.shape[0], logits.stride(0)
    targets = targets.contiguous()
    dev = logits.device
    row_loss = torch.empty(rows, dtype=F32, device=dev)
    row_lse = torch.empty(rows, dtype=F32, device=dev)
    stats = torch.empty(2, dtype=F32, device=dev)
    am = torch.empty(rows, dtype=torch.int64, device=dev) if want_argmax else None
    L.check(
        L.load().wft_ce_fwd(_p(logits), ld, _p(targets), rows, V, label_smoothing, _p(row_loss), _p(row_lse), _p(stats),
                            _p(am), L.stream_ptr()),
        "wft_ce_fwd",
    )
    return row_loss, row_lse, stats, am


def ce_bwd(logits, targets, V: int, label_smoothing: float, row_lse, stats, gscale, inplace=True):
    rows, ld = logits.shape[0], logits.stride(0)
    dl = logits if inplace else torch.empty_like(logits)
    gscale = gscale.reshape(1).to(F32)
    L.check(
        L.load().wft_ce_bwd(_p(logits), ld, _p(targets.contiguous()), rows, V, label_smoothing, _p(row_lse), _p(stats),
                            _p(gscale), _p(dl), L.stream_ptr()),
        "wft_ce_bwd",
    )
    return dl


def token_stats(logits, targets, V: int):
    """logits bf16 [rows, ld>=V]; targets i64 [rows] or None -> (stats f32 [rows, 4] = lse, max, E_p[x], x_target; argmax i64)."""
    _chk(logits, BF16, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, ld = logits.shape[0], logits.stride(0)
    out4 = torch.empty((rows, 4), dtype=F32, device=logits.device)
    am = torch.empty(rows, dtype=torch.int64, device=logits.device)
    tg = None if targets is None else targets.contiguous()
    L.check(L.load().wft_token_stats(_p(logits), ld, _p(tg), rows, V, _p(out4), _p(am), L.stream_ptr()), "wft_token_stats")
    return out4, am


# --------------------------------------------------------------------------- audio
def logmel(audio, filters, n_frames: int = 3000):
    """audio f32 [B, 160*n_frames], filters f32 [n_mels, 201] -> f32 [B, n_mels, n_frames]."""
    _chk(audio, F32, "audio"); _chk(filters, F32, "filters")
    audio = audio.contiguous(); filters = filters.contiguous()
    B, n = audio.shape
    n_mels = filters.shape[0]
    out = torch.empty((B, n_mels, n_frames), dtype=F32, device=audio.device)
    clipmax = torch.empty(B, dtype=F32, device=audio.device)
    L.check(L.load().wft_logmel(_p(audio), _p(filters), _p(out), _p(clipmax), B, n, n_mels, n_frames, L.stream_ptr()), "wft_logmel")
    return out


def specaug(mel, params, extremes=None):
    """mel f32 [B, n_mels, T]; params i32 [B, 8]; extremes i32 [B, 2] or None."""
    _chk(mel, F32, "mel"); _chk(params, torch.int32, "params")
    mel = mel.contiguous(); params = params.contiguous()
    B, n_mels, T = mel.shape
    out = torch.empty_like(mel)
    if extremes is not None:
        _chk(extremes, torch.int32, "extremes")
        extremes = extremes.contiguous()
    L.check(L.load().wft_specaug(_p(mel), _p(out), _p(params), _p(extremes), B, n_mels, T, L.stream_ptr()), "wft_specaug")
    return out


def mel_to_tmajor(mel, c_pad: int):
    _chk(mel, F32, "mel")
    mel = mel.contiguous()
    B, n_mels, T = mel.shape
    out = torch.empty((B, T + 2, c_pad), dtype=BF16, device=mel.device)
    L.check(L.load().wft_mel_to_tmajor_bf16(_p(mel), _p(out), B, n_mels, T, c_pad, L.stream_ptr()), "wft_mel_to_tmajor_bf16")
    return out


# --------------------------------------------------------------------------- optimizer
def adamw_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, bc1, bc2, gscale=None):
    L.check(
        L.load().wft_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), lr, beta1, beta2, eps, wd, bc1, bc2,
                                _p(gscale), L.stream_ptr()),
        "wft_adamw_step",
    )


def sumsq(g, out):
    L.check(L.load().wft_sumsq_f32(_p(g), g.numel(), _p(out), L.stream_ptr()), "wft_sumsq_f32")


# --------------------------------------------------------------------------- multi-tensor optimizer steps
MT_CHUNK = 65536  # include/wft.h WFT_MT_CHUNK


class _Stager:
    """Small integer tables (tensor addresses, element counts) to the device WITHOUT a host synchronisation.
    `torch.tensor(list, device=...)` goes through pageable memory: the copy waits for everything queued on the stream and the
    host waits for the copy — thirteen such stalls per optimizer step emptied the queue in front of every Muon bucket
    (profiles/r03_lora_gap_analysis.log: 16 of the 17.7 ms the GPU idled per LoRA + Muon step).  Here the values are written
    into a pinned staging buffer (recycled once its copy event has fired) and copied stream-ordered."""

    def __init__(self):
        self.pool = {}  # (dtype, capacity) -> [(pinned tensor, numpy view, event)]

    def upload(self, values, dtype, device) -> torch.Tensor:
        n = len(values)
        cap = max(256, 1 << max(n - 1, 0).bit_length())
        pool = self.pool.setdefault((dtype, cap), [])
        for i, ent in enumerate(pool):
            if ent[2].query():
                pool.pop(i)
                break
        else:
            host = torch.empty(cap, dtype=dtype, pin_memory=True)
            ent = (host, host.numpy(), torch.cuda.Event())
        host, view, ev = ent
        view[:n] = values
        dev = torch.empty(n, dtype=dtype, device=device)
        dev.copy_(host[:n], non_blocking=True)
        ev.record()
        pool.append(ent)
        return dev


_STAGER = _Stager()


def upload_table(values, dtype, device) -> torch.Tensor:
    """1-D integer table -> device tensor, stream-ordered, no host synchronisation (CPU tensors: plain construction)."""
    if torch.device(device).type != "cuda":
        return torch.tensor(values, dtype=dtype, device=device)
    return _STAGER.upload(values, dtype, device)


class TensorTable:
    """Device-side pointer table for the wft_mt_* / wft_muon_*_mt entry points: `rows` lists of equally long tensor
    lists (row r of tensor t at tab[r * n + t]).  numel / chunk_start depend only on the shapes and are cached by
    the caller; the pointer rows are rebuilt every step because gradient tensors are re-allocated."""

    def __init__(self, tensors):
        self.n = len(tensors)
        self.device = tensors[0].device
        numel = [t.numel() for t in tensors]
        starts, tot = [], 0
        for ne in numel:
            starts.append(tot)
            tot += (ne + MT_CHUNK - 1) // MT_CHUNK
        starts.append(tot)
        self.total_chunks = tot
        self.numel = upload_table(numel, torch.int64, self.device)
        self.chunk_start = upload_table(starts, torch.int32, self.device)

    def pointers(self, *rows, allow_none=False, static=(), dtypes=None):
        """`static`: indices of rows whose tensors live as long as the table (parameters, optimizer state): each is checked once
        (dtype / layout / device do not change under a live tensor object) — with 1 259 parameters the three checks on four rows
        were most of the 2.6 ms the GPU idled in front of the AdamW launch (profiles/r03_headline_gap_analysis.log)."""
        flat = []
        seen = self.__dict__.setdefault("_checked", set())
        for ri, r in enumerate(rows):
            assert len(r) == self.n
            trust = ri in static
            want = F32 if dtypes is None else dtypes[ri]  # (the 8-bit optimizer's code rows are uint8)
            for t in r:
                if t is None and allow_none:
                    flat.append(0)
                    continue
                ptr = t.data_ptr()
                # validated once per (object, storage address, dtype): an id recycled by another tensor after
                # optimizer.load_state_dict, or a parameter re-typed in place (.data = ..., model.half()), is checked again
                key = (id(t), ptr, t.dtype)
                if not (trust and key in seen):
                    if t.dtype != want or not t.is_contiguous() or not t.is_cuda:
                        raise L.WftError(f"multi-tensor optimizer kernels need contiguous {want} HIP tensors")
                    if trust:
                        seen.add(key)
                flat.append(ptr)
        return upload_table(flat, torch.int64, self.device)


def mt_sumsq(table: TensorTable, grads, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum over all tensors of g^2 -> f32 [1] on the device (fixed-order reduction); a None gradient adds nothing."""
    tab = table.pointers(grads, allow_none=True)
    partial = torch.empty(table.total_chunks, dtype=F32, device=table.device)
    out = torch.empty(1, dtype=F32, device=table.device) if out is None else out
    L.check(L.load().wft_mt_sumsq_f32(_p(tab), _p(table.numel), _p(table.chunk_start), table.n, table.total_chunks,
                                      _p(partial), _p(out), L.stream_ptr()), "wft_mt_sumsq_f32")
    return out


def mt_adamw(table: TensorTable, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, wd, bc1, bc2, sumsq=None,
             max_norm=0.0):
    tab = table.pointers(params, grads, exp_avg, exp_avg_sq, static=(0, 2, 3))
    L.check(L.load().wft_mt_adamw(_p(tab), _p(table.numel), _p(table.chunk_start), table.n, table.total_chunks, lr, beta1,
                                  beta2, eps, wd, bc1, bc2, _p(sumsq), float(max_norm), L.stream_ptr()), "wft_mt_adamw")


def mt_adamw8(table: TensorTable, params, grads, state1, state2, absmax1, absmax2, qmap1, qmap2, lr, beta1, beta2, eps, wd, bc1, bc2,
              sumsq=None, max_norm=0.0):
    """Block-wise 8-bit AdamW for every tensor of the table in one launch (wft_mt_adamw8; include/wft.h)."""
    U8 = torch.uint8
    tab = table.pointers(params, grads, state1, state2, absmax1, absmax2, static=(0, 2, 3, 4, 5), dtypes=(F32, F32, U8, U8, F32, F32))
    L.check(L.load().wft_mt_adamw8(_p(tab), _p(table.numel), _p(table.chunk_start), table.n, table.total_chunks, _p(qmap1), _p(qmap2),
                                   lr, beta1, beta2, eps, wd, bc1, bc2, _p(sumsq), float(max_norm), L.stream_ptr()), "wft_mt_adamw8")


def transpose_bf16(src: torch.Tensor, dst: torch.Tensor):
    """src bf16 [batch, rows, cols] contiguous -> dst [batch, cols, rows]."""
    _chk(src, BF16, "src"); _chk(dst, BF16, "dst")
    b, r, c = src.shape
    assert src.is_contiguous() and dst.is_contiguous() and dst.shape == (b, c, r)
    L.check(L.load().wft_transpose_bf16(_p(src), r, c, _p(dst), b, L.stream_ptr()), "wft_transpose_bf16")
    return dst


NS_COEFFS = (3.4445, -4.7750, 2.0315)  # muon.py zeropower_via_newtonschulz5


def muon_group_step(params, grads, bufs, lr, wd, momentum, nesterov=True, ns_steps=5, sumsq=None, max_norm=0.0,
                    return_update=False, shard=None, validated=False):
    """One Muon step for a list of same-shape 2-D f32 parameters (muon.py muon_update + the p update of
    SingleDeviceMuonWithAuxAdam.step): momentum/nesterov -> bf16 -> Frobenius normalisation -> 5 Newton-Schulz
    iterations as batched MFMA GEMMs -> p = p(1 - lr wd) - lr sqrt(max(1, rows/cols)) X.

    shard = (rank, world, all_gather): the distributed variant (muon.MuonWithAuxAdam, reference model/optimizer.py:227-228,
    SURVEY.md §2.2 C6).  The matrices of the group are dealt to the ranks in contiguous chunks of ceil(n / world); a rank
    runs momentum + Newton-Schulz for ITS chunk only (`grads` / `bufs` then hold that chunk: bufs[i] belongs to
    params[lo + i]), the orthogonalised updates — a bf16 quantity — are all-gathered (half the bytes of gathering the fp32
    parameters, the package's form) and every rank applies all n updates in fp32, so parameters stay bit-identical across
    ranks.  `all_gather(out, inp)` fills out [world * chunk, ...] from every rank's inp [chunk, ...]."""
    lib = L.load()
    n_all = len(params)
    rows, cols = params[0].shape
    dev = params[0].device
    tall = rows > cols
    R, Cc = (cols, rows) if tall else (rows, cols)
    Rp, Cp = round_up(R, 128), round_up(Cc, 128)
    numel = rows * cols
    chunks = (numel + MT_CHUNK - 1) // MT_CHUNK
    if shard is None:
        lo, hi, per = 0, n_all, n_all
    else:
        rank, world, all_gather = shard
        per = (n_all + world - 1) // world
        lo, hi = min(rank * per, n_all), min((rank + 1) * per, n_all)
    n = hi - lo
    if len(grads) != n or len(bufs) != n:
        raise L.WftError(f"muon_group_step: {n} owned matrices but {len(grads)} gradients / {len(bufs)} momentum buffers")
    # (validated: the caller checked `params` and `bufs` when it built them — the optimizer's per-group plan; gradients are new
    # tensors every step and are always checked)
    for row in ((grads,) if validated else (params, grads, bufs)):
        for t in row:
            if t is None and row is grads:
                continue  # no gradient this step: an all-zero one to the momentum kernel (muon.py "force synchronization")
            if t.dtype != F32 or not t.is_contiguous() or t.shape != (rows, cols) or not t.is_cuda:
                raise L.WftError("muon_group_step needs contiguous f32 HIP tensors of one shape (there is no CPU path)")
    O = None
    ldo, so = (Rp, Cp * Rp) if tall else (Cp, Rp * Cp)
    if n > 0:
        flat = [t.data_ptr() for t in params[lo:hi]] + [0 if t is None else t.data_ptr() for t in grads] + [t.data_ptr() for t in bufs]
        tab = upload_table(flat, torch.int64, dev)
        U = torch.empty((n, rows, cols), dtype=BF16, device=dev)
        partial = torch.empty((n, chunks), dtype=F32, device=dev)
        L.check(lib.wft_muon_momentum_mt(_p(tab), n, numel, momentum, int(nesterov), _p(U), _p(partial), _p(sumsq),
                                         float(max_norm), L.stream_ptr()), "wft_muon_momentum_mt")
        X = torch.empty((n, Rp, Cp), dtype=BF16, device=dev)
        Xt = torch.empty((n, Cp, Rp), dtype=BF16, device=dev)
        L.check(lib.wft_muon_prepare(_p(U), rows, cols, _p(partial), chunks, _p(X), _p(Xt), Rp, Cp, n, L.stream_ptr()),
                "wft_muon_prepare")
        del U
        A = torch.empty((n, Rp, Rp), dtype=BF16, device=dev)
        Bm = torch.empty((n, Rp, Rp), dtype=BF16, device=dev)
        Xt2 = torch.empty((n, Cp, Rp), dtype=BF16, device=dev)
        a, b, c = NS_COEFFS
        for it in range(ns_steps):
            gemm_nt(X, X, M=Rp, N=Rp, K=Cp, lda=Cp, ldb=Cp, out=A, ldc=Rp, batch=n, strideA=Rp * Cp, strideB=Rp * Cp,
                    strideC=Rp * Rp)                                                        # A = X X^T
            gemm_nt(A, A, M=Rp, N=Rp, K=Rp, lda=Rp, ldb=Rp, out=Bm, ldc=Rp, batch=n, strideA=Rp * Rp, strideB=Rp * Rp,
                    strideC=Rp * Rp, alpha=c, residual=A, ldr=Rp, strideR=Rp * Rp, beta=b)  # B = b A + c A A   (A = A^T)
            gemm_nt(Xt, Bm, M=Cp, N=Rp, K=Rp, lda=Rp, ldb=Rp, out=Xt2, ldc=Rp, batch=n, strideA=Cp * Rp, strideB=Rp * Rp,
                    strideC=Cp * Rp, residual=Xt, ldr=Rp, strideR=Cp * Rp, beta=a)          # X'^T = a X^T + X^T B (B = B^T)
            Xt, Xt2 = Xt2, Xt
            if it + 1 < ns_steps or not tall:
                transpose_bf16(Xt, X)
        O = Xt if tall else X
    if shard is not None:
        mine = torch.zeros((per,) + ((Cp, Rp) if tall else (Rp, Cp)), dtype=BF16, device=dev)
        if n > 0:
            mine[:n].copy_(O)
        O = torch.empty((world * per,) + tuple(mine.shape[1:]), dtype=BF16, device=dev)
        all_gather(O, mine)
    scale = max(1.0, rows / cols) ** 0.5
    ptab = upload_table([t.data_ptr() for t in params], torch.int64, dev)
    L.check(lib.wft_muon_apply_mt(_p(ptab), n_all, rows, cols, _p(O), ldo, so, lr, wd, scale, L.stream_ptr()), "wft_muon_apply_mt")
    if return_update:
        return O[:n_all, :rows, :cols].float() * scale
    return None
