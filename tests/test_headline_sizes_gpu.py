"""Correctness at the sizes bench.py TIMES (VERDICT r5 "weak" 1a): the headline is 87 clips of whisper-large-v3 — 130 500 activation rows,
1.34 GB GEMM operands, byte offsets close to the 32-bit eligibility limits of the one-wave-per-SIMD kernels (gemm_nt4w.hip, gemm_tn4w.hip,
attn.hip: the dispatch predicates) — and the turbo YAML's B = 64, S = 448 logits (28 672 x 51 866).  The CPU oracle cannot run these sizes:
the checks are size-independent properties (identity, linearity, rows of a softmax sum to one, rows of a CE gradient sum to zero) and
a chunked fp32 product on the GPU, each also asserting WHICH kernel the dispatcher chose.
Reference call sites: whisper.model.Linear / MultiHeadAttention.qkv_attention reached from model/model_utils.py:283-285,320-325;
F.cross_entropy at model/model_utils.py:66-68."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"
ROWS = 87 * 1500  # bench.py's headline batch


def bf(x):
    return x.to(torch.bfloat16)


def _randn_bf16(rows, cols, gen, scale=1.0):
    """bf16 N(0, scale^2) [rows, cols] without an fp32 copy of the whole tensor alive."""
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    step = 16384
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        out[r0:r0 + n] = bf(torch.randn(n, cols, device=DEV, generator=gen) * scale)
    return out


@pytest.mark.parametrize("N,Kd", [(1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)])
def test_headline_nt_gemm_identity_and_linearity(N, Kd):
    """The four encoder GEMM shapes of the headline step (out-proj, fused QKV, fc1, fc2) at M = 130 500."""
    g = torch.Generator(device=DEV).manual_seed(N + Kd)
    a1 = _randn_bf16(ROWS, Kd, g)
    args, _ = K.gemm_nt(a1, bf(torch.zeros(N, Kd, device=DEV)), _args_only=True)
    assert L.load().wft_gemm_nt_variant(C.byref(args)) == 4, "the headline shapes run on gemm_nt4w_kernel"
    eye = bf(torch.eye(N, Kd, device=DEV))
    y = K.gemm_nt(a1, eye)
    n0 = min(N, Kd)
    assert torch.equal(y[:, :n0], a1[:, :n0])  # A @ I^T == A exactly, every row of every tile
    if N > Kd:
        assert not y[:, Kd:].any()
    del y
    # rows in the LAST row tile (130 500 = 509 * 256 + 196) and bias: C = A W^T + b against an fp32 product of those rows
    w = bf(torch.randn(N, Kd, device=DEV, generator=g) * 0.05)
    bias = torch.randn(N, device=DEV, generator=g)
    y = K.gemm_nt(a1, w, bias=bias)
    for r0 in (0, 65536 - 128, ROWS - 196 - 64, ROWS - 200):
        ref = a1[r0:r0 + 200].float() @ w.float().t() + bias
        err = (y[r0:r0 + 200].float() - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item(), (r0, err)
    # linearity in A (fp32 outputs): (a1 + a2) W^T == a1 W^T + a2 W^T up to the rounding of the summed operand, which is corrected
    # to first order by the GEMM of the rounding residual
    a2 = _randn_bf16(ROWS, Kd, g)
    s = torch.empty_like(a1)
    resid = torch.empty_like(a1)
    for r0 in range(0, ROWS, 16384):
        sl = slice(r0, min(r0 + 16384, ROWS))
        exact = a1[sl].float() + a2[sl].float()
        s[sl] = bf(exact)
        resid[sl] = bf(s[sl].float() - exact)
    y1 = K.gemm_nt(a1, w, out_f32=True)
    y1 += K.gemm_nt(a2, w, out_f32=True)
    y1 += K.gemm_nt(resid, w, out_f32=True)
    r = K.gemm_nt(s, w, out_f32=True)
    assert ((r - y1).abs().max() / y1.abs().max()).item() < 2e-3


@pytest.mark.parametrize("P", [1280, 3840, 5120])
def test_headline_weight_gradient_matches_chunked_fp32(P):
    """dW = dy^T x with 130 500 reduction rows (out-proj / QKV / fc1 weight gradients of the headline step) against an fp32 product
    accumulated in 15 chunks on the GPU; bitwise reproducible across two launches."""
    g = torch.Generator(device=DEV).manual_seed(P)
    dy = _randn_bf16(ROWS, P, g)
    x = _randn_bf16(ROWS, 1280, g)
    got = K.gemm_tn(dy, x)
    assert torch.equal(got, K.gemm_tn(dy, x))
    ref = torch.zeros(P, 1280, device=DEV, dtype=torch.float64)
    for r0 in range(0, ROWS, 8700):
        ref += (dy[r0:r0 + 8700].float().t() @ x[r0:r0 + 8700].float()).double()
    err = (got.double() - ref).abs().max().item()
    assert err <= 3e-5 * ref.abs().max().item() + 1e-3, err  # fp32 accumulation over 130 500 terms of size ~1, two different orders
    # ragged reduction tail: 130 500 = 2039 * 64 + 4 rows — the last 4 rows count
    x2 = x.clone()
    x2[-4:] = 0
    d = (got - K.gemm_tn(dy, x2)).double()
    ref_tail = (dy[-4:].float().t() @ x[-4:].float()).double()
    assert (d - ref_tail).abs().max().item() <= 1e-3 + 1e-4 * ref_tail.abs().max().item()


def test_headline_encoder_attention_properties():
    """B = 87, 20 heads, 1500 x 1500 on the fused [B*T, 3d] QKV buffer layout the model uses (row stride 3840).  v = 1 makes every
    output row the sum of its softmax weights (= 1); dO = 1 then gives dS = P (dP - delta) = 0, so dq = dk = 0, and dv's column sums over
    keys are the number of queries."""
    B, H, T = 87, 20, 1500
    d = H * 64
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = _randn_bf16(B * T, 3 * d, g).view(B, T, 3 * d)
    qkv[..., 2 * d:] = 1.0
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    a = L.AttnArgs()
    a.q, a.ldq, a.q_bs = K._attn_view(q)
    a.k, a.ldk, a.k_bs = K._attn_view(k)
    a.v, a.ldv, a.v_bs = K._attn_view(v)
    a.B, a.H, a.Tq, a.Tk, a.causal, a.scale = B, H, T, T, 0, 0.125
    a.lddo, a.do_bs = d, T * d
    lib = L.load()
    assert lib.wft_attn_variant(C.byref(a), 0) == 2 and lib.wft_attn_variant(C.byref(a), 1) == 4 and lib.wft_attn_variant(C.byref(a), 2) == 4, \
        "the encoder call runs on attn_fwd_pipe_kernel / attn_bwd_dq4w_kernel / attn_bwd_dkdv4w_kernel"
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    assert (o.float() - 1).abs().max().item() < 1e-2
    # lse of the first and the last (batch, head) against fp32 math
    for b, h in ((0, 0), (B - 1, H - 1), (43, 7)):
        s = (q[b, :, 64 * h:64 * h + 64].float() @ k[b, :, 64 * h:64 * h + 64].float().t()) * 0.125
        assert (lse[b, h] - torch.logsumexp(s, -1)).abs().max().item() < 2e-3
    do = torch.ones_like(o)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
    assert dq.float().abs().max().item() < 2e-2 and dk.float().abs().max().item() < 2e-2
    assert (dv.float().view(B, T, H, 64).sum(1) - T).abs().max().item() < 0.02 * T
    del do, dq, dk, dv, o
    # a real backward on one slice of the batch, last batch element included: the 87-clip launch against the same kernels run on
    # that element alone (same arithmetic per (batch, head): bit-identical) and against fp32 math
    qkv[..., 2 * d:] = _randn_bf16(B * T, d, g).view(B, T, d)
    do = _randn_bf16(B * T, d, g).view(B, T, d)
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
    b = B - 1
    o1, lse1 = K.attn_fwd(q[b:], k[b:], v[b:], H, False, 0.125)
    dq1, dk1, dv1 = K.attn_bwd(q[b:], k[b:], v[b:], o1, lse1, do[b:], H, False, 0.125)
    assert torch.equal(o[b:], o1) and torch.equal(dq[b:], dq1) and torch.equal(dk[b:], dk1) and torch.equal(dv[b:], dv1)
    h = H - 1
    qf, kf, vf = (t[b, :, 64 * h:64 * h + 64].float().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax((qf @ kf.t()) * 0.125, -1) @ vf
    ref.backward(do[b, :, 64 * h:64 * h + 64].float())
    for got, want, nm in ((o, ref, "o"), (dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        e = (got[b, :, 64 * h:64 * h + 64].float() - want).norm() / want.norm()
        assert e.item() < 2e-2, (nm, e.item())


def test_turbo_yaml_size_ce_gradient_rows_sum_to_zero():
    """config_turbo_best.yaml's B = 64 at S = 448: 28 672 rows x 51 866 classes, 2.98 GB of bf16 logits (row stride 51 968)."""
    rows, V = 64 * 448, 51866
    ld = K.round_up(V, 128)
    g = torch.Generator(device=DEV).manual_seed(2)
    logits = _randn_bf16(rows, ld, g)
    tgt = torch.randint(0, V, (rows,), device=DEV, generator=g)
    tgt[::7] = -100
    _, lse, stats, am = K.ce_fwd(logits, tgt, V, 0.1, want_argmax=True)
    for r0 in (0, rows - 512):
        assert torch.equal(am[r0:r0 + 512], logits[r0:r0 + 512, :V].float().argmax(-1))
        assert (lse[r0:r0 + 512] - torch.logsumexp(logits[r0:r0 + 512, :V].float(), -1)).abs().max().item() < 1e-3
    dl = K.ce_bwd(logits, tgt, V, 0.1, lse, stats, torch.ones(1, device=DEV), inplace=True)
    worst = 0.0
    for r0 in range(0, rows, 2048):
        worst = max(worst, dl[r0:r0 + 2048, :V].float().sum(-1).abs().max().item())
    assert worst < 1e-4
    assert not dl[::7, :V].any()  # ignored rows carry no gradient
