"""GPU tests of the one-wave-per-SIMD dQ kernel (csrc/attn.hip, attn_bwd_dq4w_kernel), called through the C ABI.

The kernel serves the non-causal attention backward of the encoder blocks (reference: whisper's MultiHeadAttention reached through
src/whisper_finetune/model/model_utils.py:283-285; its backward is autograd's).  It forms every dQ element from the same products
in the same order as the 8-wave kernel (32-key MFMA steps, ascending keys; row constants -lse/scale and -delta as the C operand of
the first MFMA of each chain) and writes the same row constants for the dK/dV kernel, so dQ, dK, dV and both bias-gradient sums must
agree BIT FOR BIT between the variants; dQ is also checked against fp32 torch math at bf16 resolution.  Keys past Tk are switched
off element by element in the last 64-key tile only: the shapes below end that tile after 6, 13, 1, 28 and 64 valid keys.
The K loop is hand-synchronised: every case runs on fresh data more than once.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _restore_variant():
    old = K.set_variant("dq", -1)
    yield
    K.set_variant("dq", old)


def _ref_dq(q, k, v, do, H, scale):
    B, Tq, _ = q.shape
    qf, kf, vf, dof = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v, do))
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1)
    dp = dof @ vf.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    return (ds @ kf * scale).transpose(1, 2).reshape(B, Tq, H * 64)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


SHAPES = [(2, 8, 1500, 1500), (1, 8, 600, 70), (2, 8, 513, 333), (1, 5, 777, 257), (1, 8, 512, 64), (2, 20, 1500, 1500)]


@pytest.mark.parametrize("B,H,Tq,Tk", SHAPES)
def test_dq4w_matches_the_8_wave_kernel_bit_for_bit_and_fp32_math(B, H, Tq, Tk):
    lib = L.load()
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + Tq + Tk)
    D = H * 64
    for rep in range(2):
        qkv = torch.randn(B, Tq, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
        kv = torch.randn(B, Tk, 2 * D, device=DEV, generator=g).to(torch.bfloat16)
        q, k, v = qkv[..., :D], kv[..., :D], kv[..., D:]
        do = torch.randn(B, Tq, D, device=DEV, generator=g).to(torch.bfloat16)
        o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
        outs = []
        for var in (1, 0):
            K.set_variant("dq", var)
            cs = (torch.full((D,), float("nan"), device=DEV), torch.full((D,), float("nan"), device=DEV))
            dq = torch.full((B, Tq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
            _, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, dq=dq, colsums=cs)
            outs.append((dq, dk.clone(), dv.clone(), cs[0], cs[1]))
        for name, a, b in zip(("dQ", "dK", "dV", "q-bias sums", "v-bias sums"), outs[0], outs[1]):
            assert torch.equal(a, b), f"{name} differs between the two dQ kernels"
        assert _rel(outs[1][0], _ref_dq(q, k, v, do, H, 0.125)) < 4e-3
        ref_cs = outs[1][0].float().sum((0, 1))
        assert ((outs[1][3] - ref_cs).abs().max() / outs[1][0].float().abs().sum((0, 1)).max()).item() < 1e-5


def test_dq4w_dispatch_rules():
    """causal calls and calls with fewer than 512 queries keep the 8-wave kernel"""
    lib = L.load()
    g = torch.Generator(device=DEV).manual_seed(3)
    for Tq, Tk, causal in ((448, 1500, False), (640, 640, True)):
        D = 4 * 64
        q, k, v, do = (torch.randn(2, t, D, device=DEV, generator=g).to(torch.bfloat16) for t in (Tq, Tk, Tk, Tq))
        o, lse = K.attn_fwd(q, k, v, 4, causal, 0.125)
        res = []
        for var in (1, 0):
            K.set_variant("dq", var)
            res.append(K.attn_bwd(q, k, v, o, lse, do, 4, causal, 0.125))
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_dq4w_is_bitwise_reproducible_under_load():
    lib = L.load()
    K.set_variant("dq", 0)
    g = torch.Generator(device=DEV).manual_seed(5)
    B, H, T = 4, 20, 1500
    D = H * 64
    qkv = torch.randn(B, T, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(B, T, D, device=DEV, generator=g).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    first = [t.clone() for t in K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)]
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    for _ in range(8):
        with torch.cuda.stream(side):
            junk.add_(1)
        for a, b in zip(K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125), first):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
