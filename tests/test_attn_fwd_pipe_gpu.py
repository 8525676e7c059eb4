"""GPU tests of the software-pipelined forward attention kernel (csrc/attn.hip, attn_fwd_pipe_kernel), called through the C ABI.

The kernel serves the non-causal attention forward of the encoder blocks and the decoder's cross attention over long key ranges
(reference: whisper's MultiHeadAttention.qkv_attention reached through src/whisper_finetune/model/model_utils.py:283-285,320-322).
It issues the S MFMA chains of key tile kt+1 behind the softmax of tile kt (four-slot K/V ring staged three tiles ahead) and performs
the same arithmetic in the same order as attn_fwd_kernel — the stale running maximum that enters S(kt+1) as its initial accumulator is
the one tile kt's softmax has just settled — so o and lse must agree BIT FOR BIT between the two kernels; both are also checked
against fp32 torch math at bf16 resolution.  Key ranges that end inside a tile, that are shorter than the ring (1-3 tiles) and
query blocks whose last wave is empty (T = 1500) are all in the list; the loop is hand-synchronised, so every case runs on fresh
data more than once.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _restore_variant():
    old = K.set_variant("fwd", -1)
    yield
    K.set_variant("fwd", old)


def _ref(q, k, v, H, scale):
    B, Tq, _ = q.shape
    qf, kf, vf = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * scale
    return (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Tq, H * 64), torch.logsumexp(s, -1)


# (B, H, Tq, Tk): Tk >= 512 is what the dispatcher sends to the pipelined kernel by default
SHAPES = [(2, 8, 1500, 1500), (1, 8, 128, 1500), (2, 6, 50, 1500), (1, 5, 777, 513), (1, 8, 600, 512), (2, 20, 1500, 1500),
          (1, 4, 333, 577), (1, 8, 448, 1500)]


@pytest.mark.parametrize("B,H,Tq,Tk", SHAPES)
@pytest.mark.parametrize("amp", [1.0, 6.0])  # amp 6: large score spread, the lazy-rescale path fires on many tiles
def test_pipelined_forward_equals_the_unpipelined_kernel_bit_for_bit_and_fp32_math(B, H, Tq, Tk, amp):
    lib = L.load()
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + Tq + Tk)
    D = H * 64
    for rep in range(2):
        q = (torch.randn(B, Tq, D, device=DEV, generator=g) * amp).to(torch.bfloat16)
        kv = torch.randn(B, Tk, 2 * D, device=DEV, generator=g).to(torch.bfloat16)
        k, v = kv[..., :D], kv[..., D:]
        outs = []
        for var in (1, 0):
            K.set_variant("fwd", var)
            o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
            outs.append((o.clone(), lse.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (rep, "pipelined != unpipelined")
        o_ref, lse_ref = _ref(q, k, v, H, 0.125)
        assert ((outs[1][0].float() - o_ref).norm() / o_ref.norm()).item() < 6e-3
        assert (outs[1][1] - lse_ref).abs().max().item() < 2e-3 * max(1.0, lse_ref.abs().max().item())


def test_short_and_causal_calls_keep_the_unpipelined_kernel():
    """Tk < 512 and causal calls are not sent to the pipelined kernel (its longer prologue loses 2-5 % there); the switch changes
    nothing for them."""
    lib = L.load()
    torch.manual_seed(0)
    for B, H, T, causal in ((3, 6, 128, True), (2, 8, 448, True), (2, 8, 200, False)):
        qkv = torch.randn(B, T, 3 * H * 64, device=DEV).to(torch.bfloat16)
        q, k, v = qkv.chunk(3, dim=-1)
        res = []
        for var in (1, 0):
            K.set_variant("fwd", var)
            res.append(K.attn_fwd(q, k, v, H, causal, 0.125))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
