"""GPU tests of the one-wave-per-SIMD dK/dV kernel (csrc/attn.hip, attn_bwd_dkdv4w_kernel), called through the C ABI.

The kernel serves the non-causal attention backward of the encoder blocks and of the decoder's cross attention (reference:
whisper's MultiHeadAttention reached through src/whisper_finetune/model/model_utils.py:283-285, 320-325; its backward is autograd's).
It forms every dK / dV element from the same products in the same order as the 8-wave kernel (32-query MFMA steps, ascending
queries) and sums the v-bias gradient over the same lane tree, so the two must agree BIT FOR BIT; both are checked against fp32
torch math at bf16 resolution.  The K loop is hand-synchronised (counted lgkmcnt / vmcnt waits, one raw barrier per 64-query
tile): every case runs on fresh data more than once — a fragment read that runs ahead of its LDS-DMA piece returns the previous
launch's (different) operands and fails the comparison.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _restore_variant():
    old = K.set_variant("dkdv", -1)
    yield
    K.set_variant("dkdv", old)


def _ref_bwd(q, k, v, do, H, scale):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qf, kf, vf, dof = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v, do))
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1)
    dv = p.transpose(-1, -2) @ dof
    dp = dof @ vf.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    dk = ds.transpose(-1, -2) @ qf * scale
    return dk.transpose(1, 2).reshape(B, Tk, H * 64), dv.transpose(1, 2).reshape(B, Tk, H * 64)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


# (B, H, Tq, Tk): the encoder shape, cross attention (short ragged Tq, 1500 keys), ragged both ways, fewer keys than one
# workgroup's 256, Tq at the kernel's minimum of 128, a head count that is not a multiple of 8 (no XCD remap)
SHAPES = [(2, 8, 1500, 1500), (3, 6, 200, 1500), (1, 8, 128, 70), (2, 8, 449, 333), (1, 5, 321, 257), (2, 20, 1500, 1500)]


@pytest.mark.parametrize("B,H,Tq,Tk", SHAPES)
def test_dkdv4w_matches_the_8_wave_kernel_bit_for_bit_and_fp32_math(B, H, Tq, Tk):
    lib = L.load()
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + Tq + Tk)
    D = H * 64
    for rep in range(2):
        # packed projections, as the model lays them out: q inside [B, Tq, 3D], k / v inside [B, Tk, 2D] (row strides != D)
        qkv = torch.randn(B, Tq, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
        kv = torch.randn(B, Tk, 2 * D, device=DEV, generator=g).to(torch.bfloat16)
        q, k, v = qkv[..., :D], kv[..., :D], kv[..., D:]
        do = torch.randn(B, Tq, D, device=DEV, generator=g).to(torch.bfloat16)
        o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
        outs = []
        for var in (1, 0):
            K.set_variant("dkdv", var)
            cs = (torch.full((D,), float("nan"), device=DEV), torch.full((D,), float("nan"), device=DEV))
            dk = torch.full((B, Tk, D), float("nan"), dtype=torch.bfloat16, device=DEV)
            dv = torch.full((B, Tk, D), float("nan"), dtype=torch.bfloat16, device=DEV)
            K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, dk=dk, dv=dv, colsums=cs)
            outs.append((dk, dv, cs[1]))
        assert torch.equal(outs[0][0], outs[1][0]), "dK differs between the two kernels"
        assert torch.equal(outs[0][1], outs[1][1]), "dV differs between the two kernels"
        assert torch.equal(outs[0][2], outs[1][2]), "v-bias gradient differs between the two kernels"
        rk, rv = _ref_bwd(q, k, v, do, H, 0.125)
        assert _rel(outs[1][0], rk) < 4e-3 and _rel(outs[1][1], rv) < 4e-3
        assert ((outs[1][2] - outs[1][1].float().sum((0, 1))).abs().max() / outs[1][1].float().abs().sum((0, 1)).max()).item() < 1e-5


def test_dkdv4w_dispatch_rules():
    """causal calls and calls with fewer than 128 queries keep the 8-wave kernel (results equal either way: same variant runs)"""
    lib = L.load()
    g = torch.Generator(device=DEV).manual_seed(3)
    for Tq, Tk, causal in ((96, 400, False), (256, 256, True)):
        D = 4 * 64
        q, k, v, do = (torch.randn(2, t, D, device=DEV, generator=g).to(torch.bfloat16) for t in (Tq, Tk, Tk, Tq))
        o, lse = K.attn_fwd(q, k, v, 4, causal, 0.125)
        res = []
        for var in (1, 0):
            K.set_variant("dkdv", var)
            res.append(K.attn_bwd(q, k, v, o, lse, do, 4, causal, 0.125))
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_dkdv4w_is_bitwise_reproducible_under_load():
    lib = L.load()
    K.set_variant("dkdv", 0)
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
