"""The fp32 compute mode of the engine: the same Whisper forward / backward as engine/ops.py, every tensor fp32 and every
product an fp32 MFMA GEMM (csrc/f32.hip).

The reference computes in true fp32 when `training.mixed_precision_training` is False (autocast disabled:
model/model_utils.py:37-48,64; eval/evaluator.py:69) and BASELINE.json's north star asks for parity with the reference's
fp32 path within 1e-3 relative.  `whisper_model.Whisper` switches to these functions when its `compute_dtype` is "fp32"
(set from the YAML by scripts/finetune.py, or `model.set_compute_dtype("fp32")`).  Nothing is fused here: one kernel per
op of the restated upstream arithmetic, fixed summation orders (bitwise reproducible).  It is the
PARITY mode for the small configurations (BASELINE configs[0]); the throughput path is the bf16 one.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L

F32 = torch.float32


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise L.WftError(f"{name}: the fp32 mode runs on the GPU (libwft f32 kernels); got a CPU tensor")
    if t.dtype != F32:
        raise L.WftError(f"{name}: expected float32, got {t.dtype}")
    return t


def gemm(a: torch.Tensor, b: torch.Tensor, *, M: int, N: int, K: int, a_strides, b_strides, out: Optional[torch.Tensor] = None,
         ldc: Optional[int] = None, c_bs: int = 0, batch: int = 1, bias: Optional[torch.Tensor] = None, alpha: float = 1.0,
         beta: float = 0.0) -> torch.Tensor:
    """C[b][m, n] = alpha * sum_k A(b; m, k) B(b; k, n) (+ beta C) (+ bias[n]).  a_strides = (row, k, batch) element strides
    of A(m, k); b_strides = (k, n, batch) of B(k, n)."""
    _f32(a, "A"); _f32(b, "B")
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=F32, device=a.device)
        ldc, c_bs = N, M * N
        beta = 0.0
    args = L.GemmF32Args()
    args.A, (args.a_rs, args.a_cs, args.a_bs) = a.data_ptr(), a_strides
    args.B, (args.b_rs, args.b_cs, args.b_bs) = b.data_ptr(), b_strides
    args.C, args.ldc, args.c_bs = out.data_ptr(), ldc, c_bs
    args.bias = None if bias is None else _f32(bias, "bias").data_ptr()
    args.M, args.N, args.K, args.batch = M, N, K, batch
    args.alpha, args.beta = alpha, beta
    L.check(L.load().wft_gemm_f32(C.byref(args), L.stream_ptr()), "wft_gemm_f32")
    return out


def _mm_nt(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """x [M, K] @ w [N, K]^T (+ bias)."""
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, w, M=M, N=N, K=K, a_strides=(x.stride(0), x.stride(1), 0), b_strides=(w.stride(1), w.stride(0), 0), bias=bias)


def _mm_nn(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [M, K] @ w [K, N]."""
    M, K = x.shape
    N = w.shape[1]
    return gemm(x, w, M=M, N=N, K=K, a_strides=(x.stride(0), x.stride(1), 0), b_strides=(w.stride(0), w.stride(1), 0))


def _mm_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [R, P]^T @ b [R, Q] -> [P, Q] (weight gradients)."""
    R, P = a.shape
    Q = b.shape[1]
    return gemm(a, b, M=P, N=Q, K=R, a_strides=(a.stride(1), a.stride(0), 0), b_strides=(b.stride(0), b.stride(1), 0))


def colsum(x: torch.Tensor) -> torch.Tensor:
    x = _f32(x, "x")
    out = torch.empty(x.shape[1], dtype=F32, device=x.device)
    L.check(L.load().wft_colsum_f32(_p(x), x.shape[0], x.shape[1], x.stride(0), _p(out), L.stream_ptr()), "wft_colsum_f32")
    return out


def axpby(a: float, x: torch.Tensor, b: float = 0.0, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _f32(x, "x").contiguous()
    if y is not None:
        y = _f32(y, "y").contiguous()
    out = torch.empty_like(x)
    L.check(L.load().wft_axpby_f32(a, _p(x), b, _p(y), _p(out), x.numel(), L.stream_ptr()), "wft_axpby_f32")
    return out


# ------------------------------------------------------------------------------------------------ Linear (+ LoRA)
class LinearFn(torch.autograd.Function):
    """y = x W_eff^T + b, W_eff = W + s B (A * mask) (minLoRA's parametrization form) in fp32.
    The adapter gradients go through dW_eff exactly as autograd's do for the reference's parametrization."""

    @staticmethod
    def forward(ctx, x, w, b, lora_a, lora_b, scaling, mask):
        x2 = _f32(x, "x").reshape(-1, x.shape[-1]).contiguous()
        w_eff = w
        if lora_a is not None:
            am = lora_a if mask is None else lora_a * mask
            w_eff = gemm(lora_b, am, M=w.shape[0], N=w.shape[1], K=lora_a.shape[0],
                         a_strides=(lora_b.stride(0), lora_b.stride(1), 0), b_strides=(am.stride(0), am.stride(1), 0),
                         out=w.detach().clone(), ldc=w.shape[1], alpha=float(scaling), beta=1.0)
        y = _mm_nt(x2, w_eff, None if b is None else b.detach())
        ctx.save_for_backward(x2, w_eff if lora_a is not None else w, lora_a, lora_b, mask)
        ctx.cfg = (x.shape, scaling, b is not None)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w_eff, lora_a, lora_b, mask = ctx.saved_tensors
        shape, scaling, has_bias = ctx.cfg
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        need = ctx.needs_input_grad
        dx = _mm_nn(dy2, w_eff).view(shape) if need[0] else None
        dw_eff = _mm_tn(dy2, x2) if (need[1] or need[3] or need[4]) else None
        db = colsum(dy2) if (has_bias and need[2]) else None
        da = dbm = None
        if lora_a is not None and (need[3] or need[4]):
            am = lora_a if mask is None else lora_a * mask
            if need[4]:  # dB = s dW_eff (A*m)^T
                dbm = gemm(dw_eff, am, M=dw_eff.shape[0], N=am.shape[0], K=am.shape[1], a_strides=(dw_eff.stride(0), 1, 0),
                           b_strides=(am.stride(1), am.stride(0), 0), alpha=float(scaling))
            if need[3]:  # dA = s (B^T dW_eff) * m
                da = gemm(lora_b, dw_eff, M=lora_b.shape[1], N=dw_eff.shape[1], K=lora_b.shape[0],
                          a_strides=(lora_b.stride(1), lora_b.stride(0), 0), b_strides=(dw_eff.stride(0), 1, 0), alpha=float(scaling))
                if mask is not None:
                    da = da * mask
        return dx, (dw_eff if need[1] else None), db, da, dbm, None, None


def linear(x, w, b, spec=None):
    if spec is None:
        return LinearFn.apply(x, w, b, None, None, 1.0, None)
    mask = spec.mask
    if mask is not None and getattr(spec.owner, "_pool", None) is not None:
        mask = mask.clone()  # a slice of the model's persistent mask pool: autograd must save THIS forward's values (parity mode: one small copy per Linear)
    return LinearFn.apply(x, w, b, spec.A, spec.B, spec.scaling, mask)


# ------------------------------------------------------------------------------------------------ LayerNorm / GELU
def _mask_arr(mask):
    return None if mask is None else (C.c_int32 * 5)(*[int(v) for v in mask])


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, mask):
        shape = x.shape
        x2 = _f32(x, "x").reshape(-1, shape[-1]).contiguous()
        rows, cols = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=F32, device=x.device)
        rstd = torch.empty(rows, dtype=F32, device=x.device)
        L.check(L.load().wft_layernorm_fwd_f32(_p(x2), _p(gamma.detach()), _p(beta.detach()), _p(y), _p(mean), _p(rstd), rows, cols,
                                               float(eps), _mask_arr(mask), L.stream_ptr()), "wft_layernorm_fwd_f32")
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.cfg = (shape, mask)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        shape, mask = ctx.cfg
        dy2 = dy.reshape(-1, shape[-1]).contiguous()
        rows, cols = x2.shape
        dx = torch.empty_like(x2)
        dg = torch.empty(cols, dtype=F32, device=dy.device)
        db = torch.empty(cols, dtype=F32, device=dy.device)
        L.check(L.load().wft_layernorm_bwd_f32(_p(dy2), _p(x2), _p(gamma.detach()), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, cols,
                                               _mask_arr(mask), L.stream_ptr()), "wft_layernorm_bwd_f32")
        return dx.view(shape), dg, db, None, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "x").contiguous()
        y = torch.empty_like(x)
        L.check(L.load().wft_gelu_fwd_f32(_p(x), _p(y), x.numel(), L.stream_ptr()), "wft_gelu_fwd_f32")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().wft_gelu_bwd_f32(_p(dy), _p(x), _p(dx), x.numel(), L.stream_ptr()), "wft_gelu_bwd_f32")
        return dx


class AddFn(torch.autograd.Function):
    """a * x + b * y (residual adds, the stochastic-depth rescale x + (out - x) / keep)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        ctx.cfg = (a, b)
        return axpby(a, x, b, y)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.cfg
        g = g.contiguous()
        return (g if a == 1.0 else axpby(a, g)), (g if b == 1.0 else axpby(b, g)), None, None


def add(x, y):
    return AddFn.apply(x, y, 1.0, 1.0)


def sd_rescale(x, out, keep: float):
    s = 1.0 / keep
    return AddFn.apply(x, out, 1.0 - s, s)


# ------------------------------------------------------------------------------------------------ attention
class AttentionFn(torch.autograd.Function):
    """qkv_attention in fp32: softmax(q k^T / sqrt(64) (+ causal mask)) v per head, as strided batched
    GEMMs over the [B, T, H*64] projections (no head transposes) with the probabilities materialised [B*H, Tq, Tk]."""

    @staticmethod
    def forward(ctx, q, k, v, n_head, causal):
        B, Tq, D = q.shape
        Tk = k.shape[1]
        dh = D // n_head
        scale = dh ** -0.5
        dev = q.device
        p = torch.empty((B * n_head, Tq, Tk), dtype=F32, device=dev)
        o = torch.empty((B, Tq, D), dtype=F32, device=dev)
        lib = L.load()
        for b in range(B):  # batch index of the GEMM = head (stride dh inside a row); one launch per clip
            gemm(q[b], k[b], M=Tq, N=Tk, K=dh, a_strides=(q.stride(1), 1, dh), b_strides=(1, k.stride(1), dh),
                 out=p[b * n_head:(b + 1) * n_head], ldc=Tk, c_bs=Tq * Tk, batch=n_head)
        L.check(lib.wft_softmax_fwd_f32(_p(p), B * n_head * Tq, Tk, Tk, scale, int(causal), Tq, L.stream_ptr()), "wft_softmax_fwd_f32")
        for b in range(B):
            gemm(p[b * n_head:(b + 1) * n_head], v[b], M=Tq, N=dh, K=Tk, a_strides=(Tk, 1, Tq * Tk), b_strides=(v.stride(1), 1, dh),
                 out=o[b], ldc=D, c_bs=dh, batch=n_head)
        ctx.save_for_backward(q, k, v, p)
        ctx.cfg = (n_head, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        n_head, scale = ctx.cfg
        B, Tq, D = q.shape
        Tk = k.shape[1]
        dh = D // n_head
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dp = torch.empty_like(p)
        lib = L.load()
        for b in range(B):
            ph = p[b * n_head:(b + 1) * n_head]
            # dV[h] = P^T dO ;  dP = dO V^T
            gemm(ph, do[b], M=Tk, N=dh, K=Tq, a_strides=(1, Tk, Tq * Tk), b_strides=(D, 1, dh), out=dv[b], ldc=dv.stride(1), c_bs=dh,
                 batch=n_head)
            gemm(do[b], v[b], M=Tq, N=Tk, K=dh, a_strides=(D, 1, dh), b_strides=(1, v.stride(1), dh),
                 out=dp[b * n_head:(b + 1) * n_head], ldc=Tk, c_bs=Tq * Tk, batch=n_head)
        L.check(lib.wft_softmax_bwd_f32(_p(p), _p(dp), B * n_head * Tq, Tk, Tk, scale, L.stream_ptr()), "wft_softmax_bwd_f32")
        for b in range(B):
            ds = dp[b * n_head:(b + 1) * n_head]
            gemm(ds, k[b], M=Tq, N=dh, K=Tk, a_strides=(Tk, 1, Tq * Tk), b_strides=(k.stride(1), 1, dh), out=dq[b], ldc=dq.stride(1),
                 c_bs=dh, batch=n_head)
            gemm(ds, q[b], M=Tk, N=dh, K=Tq, a_strides=(1, Tk, Tq * Tk), b_strides=(q.stride(1), 1, dh), out=dk[b], ldc=dk.stride(1),
                 c_bs=dh, batch=n_head)
        return dq, dk, dv, None, None


# ------------------------------------------------------------------------------------------------ conv stem
class ConvStemFn(torch.autograd.Function):
    """gelu(conv1(mel)) -> gelu(conv2(.)) -> permute -> + positional_embedding (model/model_utils.py:276-281) in fp32:
    time-major zero-haloed buffers, a k=3 conv = one GEMM whose A rows are overlapping windows (row stride C or 2C)."""

    @staticmethod
    def forward(ctx, mel, w1, b1, w2, b2, pos):
        B, n_mels, T = mel.shape
        d = w1.shape[0]
        dev = mel.device
        T2 = T // 2
        x0 = torch.zeros((B, T + 2, n_mels), dtype=F32, device=dev)
        x0[:, 1:T + 1] = _f32(mel, "mel").transpose(1, 2)
        w1k = w1.detach().permute(0, 2, 1).reshape(d, 3 * n_mels).contiguous()   # [co, kk*C + ci]
        w2k = w2.detach().permute(0, 2, 1).reshape(d, 3 * d).contiguous()
        pre1 = gemm(x0, w1k, M=T, N=d, K=3 * n_mels, a_strides=(n_mels, 1, (T + 2) * n_mels), b_strides=(1, 3 * n_mels, 0), batch=B,
                    bias=b1.detach(), out=torch.empty((B, T, d), dtype=F32, device=dev), ldc=d, c_bs=T * d)
        act1 = torch.zeros((B, T + 2, d), dtype=F32, device=dev)
        act1[:, 1:T + 1] = GeluFn.apply(pre1)
        pre2 = gemm(act1, w2k, M=T2, N=d, K=3 * d, a_strides=(2 * d, 1, (T + 2) * d), b_strides=(1, 3 * d, 0), batch=B, bias=b2.detach(),
                    out=torch.empty((B, T2, d), dtype=F32, device=dev), ldc=d, c_bs=T2 * d)
        g2 = GeluFn.apply(pre2)
        posc = pos.detach().contiguous()
        x = torch.empty_like(g2)
        for b in range(B):  # + positional_embedding, clip by clip (one [T2, d] table for the whole batch)
            L.check(L.load().wft_axpby_f32(1.0, _p(g2[b]), 1.0, _p(posc), _p(x[b]), g2[b].numel(), L.stream_ptr()), "wft_axpby_f32")
        ctx.save_for_backward(x0, pre1, act1, pre2, w2k)
        ctx.shapes = (B, n_mels, T, d)
        return x

    @staticmethod
    def backward(ctx, dx):
        x0, pre1, act1, pre2, w2k = ctx.saved_tensors
        B, n_mels, T, d = ctx.shapes
        T2 = T // 2
        dev = dx.device
        lib = L.load()

        def dgelu(dy, pre):
            out = torch.empty_like(pre)
            L.check(lib.wft_gelu_bwd_f32(_p(dy.contiguous()), _p(pre), _p(out), pre.numel(), L.stream_ptr()), "wft_gelu_bwd_f32")
            return out

        dpre2 = dgelu(dx, pre2)                                              # [B, T2, d]
        db2 = colsum(dpre2.view(B * T2, d))
        # dW2[co, kk*d + ci] = sum_{b,t'} dpre2[b, t', co] * act1[b, 2t' + kk, ci]: the batch is an extra reduction (beta = 1)
        dw2k = torch.zeros((d, 3 * d), dtype=F32, device=dev)
        for b in range(B):
            gemm(dpre2[b], act1[b], M=d, N=3 * d, K=T2, a_strides=(1, d, 0), b_strides=(2 * d, 1, 0), out=dw2k, ldc=3 * d, beta=1.0)
        # backward-data into the padded conv1 activation: dact1[b, 2t' + kk, ci] += dpre2[b, t', co] * w2k[co, kk*d + ci]
        dact1 = torch.zeros((B, T + 2, d), dtype=F32, device=dev)
        for kk in range(3):
            gemm(dpre2, w2k[:, kk * d:(kk + 1) * d], M=T2, N=d, K=d, a_strides=(d, 1, T2 * d), b_strides=(3 * d, 1, 0), batch=B,
                 out=dact1[:, kk:], ldc=2 * d, c_bs=(T + 2) * d, beta=1.0)
        dpre1 = dgelu(dact1[:, 1:T + 1].contiguous(), pre1)                  # [B, T, d]
        db1 = colsum(dpre1.view(B * T, d))
        dw1k = torch.zeros((d, 3 * n_mels), dtype=F32, device=dev)
        for b in range(B):
            gemm(dpre1[b], x0[b], M=d, N=3 * n_mels, K=T, a_strides=(1, d, 0), b_strides=(n_mels, 1, 0), out=dw1k, ldc=3 * n_mels, beta=1.0)
        dw1 = dw1k.view(d, 3, n_mels).permute(0, 2, 1)
        dw2 = dw2k.view(d, 3, d).permute(0, 2, 1)
        return None, dw1, db1, dw2, db2, None


# ------------------------------------------------------------------------------------------------ embedding / logits / loss
class EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, emb, pos):
        B, S = tokens.shape
        d = emb.shape[1]
        out = torch.empty((B, S, d), dtype=F32, device=emb.device)
        L.check(L.load().wft_embed_fwd_f32(_p(tokens.contiguous()), _p(emb.detach()), _p(pos.detach()), _p(out), B, S, d, L.stream_ptr()),
                "wft_embed_fwd_f32")
        ctx.save_for_backward(tokens)
        ctx.shapes = (emb.shape, pos.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        es, ps = ctx.shapes
        B, S = tokens.shape
        demb = torch.zeros(es, dtype=F32, device=dout.device)
        dpos = torch.zeros(ps, dtype=F32, device=dout.device)
        L.check(L.load().wft_embed_bwd_f32(_p(tokens.contiguous()), _p(dout.contiguous()), _p(demb), _p(dpos), B, S, es[1], L.stream_ptr()),
                "wft_embed_bwd_f32")
        return None, demb, dpos


class TiedLogitsFn(torch.autograd.Function):
    """logits = x @ E^T (model/model_utils.py:325), fp32 [M, V]."""

    @staticmethod
    def forward(ctx, x, emb):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2, emb)
        ctx.shape = x.shape
        return _mm_nt(x2, emb.detach())

    @staticmethod
    def backward(ctx, dl):
        x2, emb = ctx.saved_tensors
        dl = dl.contiguous()
        dx = _mm_nn(dl, emb.detach()).view(ctx.shape) if ctx.needs_input_grad[0] else None
        de = _mm_tn(dl, x2) if ctx.needs_input_grad[1] else None
        return dx, de


class CrossEntropyFn(torch.autograd.Function):
    """mean label-smoothed CE over non-ignored targets on fp32 logits [M, V]; the backward overwrites the logits buffer."""

    @staticmethod
    def forward(ctx, logits, targets, eps):
        M, V = logits.shape
        dev = logits.device
        row_loss = torch.empty(M, dtype=F32, device=dev)
        row_lse = torch.empty(M, dtype=F32, device=dev)
        stats = torch.empty(2, dtype=F32, device=dev)
        L.check(L.load().wft_ce_fwd_f32(_p(logits), logits.stride(0), _p(targets), M, V, float(eps), _p(row_loss), _p(row_lse), _p(stats),
                                        L.stream_ptr()), "wft_ce_fwd_f32")
        ctx.save_for_backward(logits, targets, row_lse, stats)
        ctx.eps = float(eps)
        return stats[0] / stats[1]

    @staticmethod
    def backward(ctx, g):
        logits, targets, row_lse, stats = ctx.saved_tensors
        M, V = logits.shape
        gs = g.reshape(1).to(F32).contiguous()
        L.check(L.load().wft_ce_bwd_f32(_p(logits), logits.stride(0), _p(targets), M, V, ctx.eps, _p(row_lse), _p(stats), _p(gs),
                                        L.stream_ptr()), "wft_ce_bwd_f32")
        return logits, None, None
