"""GPU tests of the one-wave-per-SIMD NT GEMM (csrc/gemm_nt4w.hip), called through the C ABI.

The kernel replaces the matmul of whisper.model.Linear on the encoder-sized problems (reference:
src/whisper_finetune/model/model_utils.py:283-285, 320-325 reach it through upstream `Linear`).  It accumulates every output
element in the same k order as the 8-wave ping-pong kernel (32-deep MFMA steps, ascending k), so the two must agree BIT FOR BIT;
both are checked against fp32 torch math at bf16 resolution.  The K loop is hand-synchronised (counted vmcnt waits, raw
barriers): every case runs several times on FRESH data — a fragment read that runs ahead of its LDS-DMA piece returns the previous
launch's (different) operands and fails the comparison.
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
    return K.set_variant("nt", v)


@pytest.fixture(autouse=True)
def _restore_variant():
    old = _variant(-1)
    yield
    _variant(old)


def _is_4w(a, b, **kw):
    """the dispatcher's own answer for these arguments under variant 0 (4 = gemm_nt4w_kernel)"""
    _variant(0)  # (the variant is a field of the argument struct now: set before the arguments are built)
    args, _ = K.gemm_nt(a, b, _args_only=True, **{k: v for k, v in kw.items() if k != "colsum"})
    return L.load().wft_gemm_nt_variant(__import__("ctypes").byref(args)) == 4


def _gelu(x):
    return torch.nn.functional.gelu(x)


def _dgelu(x):
    x = x.double()
    return (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5).float()


def _rel(got, ref):
    return ((got.float() - ref).norm() / ref.norm()).item()


# 128+ tiles of 256x256 (the dispatch threshold), ragged M, several tiles per workgroup, K at the kernel's minimum and beyond
SHAPES = [(8192, 1024, 768), (8192 + 112, 1024, 896), (4500, 3840, 1280), (9000, 1280, 5120), (70000, 512, 768)]


@pytest.mark.parametrize("M,N,Kd", SHAPES)
def test_nt4w_matches_the_pingpong_kernel_bit_for_bit_and_fp32_math(M, N, Kd):
    g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
    for rep in range(3):  # fresh operands every time (see the module docstring)
        a = bf(torch.randn(M, Kd, device=DEV, generator=g))
        b = bf(torch.randn(N, Kd, device=DEV, generator=g) * 0.05)
        bias = torch.randn(N, device=DEV, generator=g)
        res = bf(torch.randn(M, N, device=DEV, generator=g))
        aux_in = bf(torch.randn(M, N, device=DEV, generator=g))
        ref0 = a.float() @ b.float().t()
        cases = [
            ("plain", dict(), ref0),
            ("bias", dict(bias=bias), ref0 + bias),
            ("bias+residual", dict(bias=bias, residual=res), ref0 + bias + res.float()),
            ("alpha, beta", dict(alpha=0.5, residual=res, beta=2.0), 0.5 * ref0 + 2.0 * res.float()),
            ("mul_aux", dict(epilogue=L.EPI_MUL_AUX, aux=aux_in), ref0 * aux_in.float()),
        ]
        for name, kw, ref in cases:
            assert _is_4w(a, b, **kw)
            _variant(0); o4 = K.gemm_nt(a, b, **kw)
            _variant(1); op = K.gemm_nt(a, b, **kw)
            assert torch.equal(o4, op), f"{name}: the two 256x256 kernels differ (max {(o4.float() - op.float()).abs().max().item():.3e})"
            assert _rel(o4, ref) < 4e-3, name
        # GELU_GRAD: C = gelu(acc + bias), aux <- gelu'(acc + bias)
        pre = ref0 + bias
        outs = []
        for v in (0, 1):
            _variant(v)
            aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            outs.append((K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux), aux))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert _rel(outs[0][0], _gelu(pre)) < 4e-3 and _rel(outs[0][1], _dgelu(pre)) < 4e-3


@pytest.mark.parametrize("M,N,Kd", [(8192 + 40, 1024, 768), (4500, 3840, 1280)])
def test_nt4w_fused_column_sums(M, N, Kd):
    """colsum[n] = sum_m C[m, n] formed in the epilogue (the bias gradient of fc1 from the fc2 backward-data product,
    engine/ops.py MLP backward); rows beyond M of a ragged last tile must not contribute."""
    g = torch.Generator(device=DEV).manual_seed(7)
    a = bf(torch.randn(M, Kd, device=DEV, generator=g)); b = bf(torch.randn(N, Kd, device=DEV, generator=g) * 0.05)
    aux = bf(torch.randn(M, N, device=DEV, generator=g))
    ref = ((a.float() @ b.float().t()) * aux.float()).sum(0)  # (the fused sums are taken before C is rounded to bf16)
    got = []
    for v in (0, 1):
        _variant(v)
        cs = torch.full((N,), float("nan"), device=DEV)
        K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=cs)
        assert ((cs - ref).norm() / ref.norm()).item() < 1e-4, v
        got.append(cs)
    _variant(0)
    cs2 = torch.empty(N, device=DEV)
    K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=cs2)
    assert torch.equal(got[0], cs2)  # fixed summation order: bitwise reproducible


def test_nt4w_batched_and_strided_operands():
    g = torch.Generator(device=DEV).manual_seed(11)
    B, M, N, Kd = 3, 4096, 1024, 768
    a = bf(torch.randn(B * M, Kd + 64, device=DEV, generator=g))[:, :Kd]  # lda > K
    b = bf(torch.randn(N, Kd, device=DEV, generator=g) * 0.05)
    out4 = torch.empty(B * M, N + 256, dtype=torch.bfloat16, device=DEV)[:, :N]  # ldc > N
    outp = torch.empty(B * M, N + 256, dtype=torch.bfloat16, device=DEV)[:, :N]  # (empty_like would drop the row stride)
    kw = dict(M=M, N=N, K=Kd, batch=B, strideA=M * (Kd + 64), strideB=0, lda=Kd + 64, ldc=N + 256, strideC=M * (N + 256))
    _variant(0); K.gemm_nt(a, b, out=out4, **kw)
    _variant(1); K.gemm_nt(a, b, out=outp, **kw)
    ref = a.float() @ b.float().t()
    assert torch.equal(out4, outp) and _rel(out4, ref) < 4e-3


def test_nt4w_is_bitwise_reproducible_under_load():
    """the same launch repeated back to back while another stream keeps the memory system busy"""
    g = torch.Generator(device=DEV).manual_seed(5)
    M, N, Kd = 20000, 1280, 1280
    a = bf(torch.randn(M, Kd, device=DEV, generator=g)); b = bf(torch.randn(N, Kd, device=DEV, generator=g) * 0.05)
    bias = torch.randn(N, device=DEV, generator=g)
    _variant(0)
    first = K.gemm_nt(a, b, bias=bias).clone()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    for _ in range(10):
        with torch.cuda.stream(side):
            junk.add_(1)
        assert torch.equal(K.gemm_nt(a, b, bias=bias), first)
    torch.cuda.synchronize()
