"""GPU tests of the one-wave-per-SIMD weight-gradient GEMM (csrc/gemm_tn4w.hip), called through the C ABI.

dW = dY^T X of whisper.model.Linear (reference: loss.backward() in src/whisper_finetune/model/model_utils.py:83-84 reaches it
through autograd).  fp32 output from bf16 operands: compared with fp32 torch math at 2e-5 (relative L2; measured 3e-7 .. 1e-6),
with the 8-wave kernel (same 32-row MFMA steps, possibly another split-K plan: agreement to fp32 rounding), and with itself bit
for bit (the split-K partials are summed in a fixed order).  Reduction lengths that are not multiples of 64 exercise the
descriptor bounds (rows beyond R must land in LDS as zeros); every case runs on fresh data twice (a fragment read ahead of its
LDS-DMA piece would return the previous launch's operands).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


def _variant(v):
    return K.set_variant("tn", v)


@pytest.fixture(autouse=True)
def _restore_variant():
    old = _variant(-1)
    yield
    _variant(old)


def _rel(got, ref):
    return ((got.float() - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("R,P,Q", [(16384, 256, 256), (20000 + 37, 512, 256), (48000, 1280, 1280), (30000 + 63, 3840, 1280), (51000, 1280, 5120), (6400, 5120, 1280), (1024 + 1, 1280, 3840)])
def test_tn4w_matches_fp32_math_and_the_pingpong_kernel(R, P, Q):
    g = torch.Generator(device=DEV).manual_seed(R + P + Q)
    for rep in range(2):
        a = bf(torch.randn(R, P, device=DEV, generator=g)); b = bf(torch.randn(R, Q, device=DEV, generator=g))
        ref = a.float().t() @ b.float()
        _variant(0); o4 = K.gemm_tn(a, b)
        _variant(1); op = K.gemm_tn(a, b)
        assert _rel(o4, ref) < 2e-5 and _rel(op, ref) < 2e-5
        assert _rel(o4, op.float()) < 2e-6
        _variant(0)
        again = K.gemm_tn(a, b)
        assert torch.equal(o4, again)  # fixed summation order
        acc = o4.clone()
        K.gemm_tn(a, b, out=acc, accumulate=True, alpha=0.5)
        assert _rel(acc, 1.5 * ref) < 2e-5


def test_tn4w_strided_operands():
    g = torch.Generator(device=DEV).manual_seed(3)
    R, P, Q = 40000, 1280, 1280
    a = bf(torch.randn(R, 3 * P, device=DEV, generator=g))[:, P:2 * P]  # the k slice of a fused qkv gradient: lda = 3 P
    b = bf(torch.randn(R, Q + 128, device=DEV, generator=g))[:, :Q]
    out = torch.empty(P, Q + 256, device=DEV)[:, :Q]
    _variant(0)
    K.gemm_tn(a, b, out=out)
    assert _rel(out, a.float().t() @ b.float()) < 2e-5
