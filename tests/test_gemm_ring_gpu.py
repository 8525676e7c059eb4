"""GPU tests of the low-occupancy forms of the 128-tile GEMMs (round 6): the four-buffer ring of gemm_nt_kernel / gemm_tn_kernel taken
by grids of at most one workgroup per CU (a decoder block's Linears at B*S = 1 024 rows), and the split-K form of the NT kernel for the
tied-embedding backward-data product (dX = dlogits E: whisper.model.TextDecoder.forward's logits matmul, reached from
/root/reference/src/whisper_finetune/model/model_utils.py:83-84).  The ring forms run the same products in the same order as the
two-buffer kernels: bit-identical; split-K partials are summed in split order: reproducible, fp32-exact up to the summation order."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,Kd", [(1000, 384, 512), (1024, 384, 2048), (130, 128, 192), (3000, 640, 64)])
def test_ring_nt_is_bit_identical_to_the_two_buffer_kernel(M, N, Kd):
    """The same rows computed inside a problem of > 256 tiles (two-buffer kernel; N % 256 != 0 keeps the 256 x 256 kernels out) and alone
    (ring): every epilogue, bit for bit."""
    g = torch.Generator().manual_seed(M + N + Kd)
    a = bf(torch.randn(M, Kd, generator=g)).to(DEV); b = bf(torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV); res = bf(torch.randn(M, N, generator=g)).to(DEV)
    rep = (256 * 128) // (M * (N // 128)) + 2
    A = a.repeat(rep, 1); R = res.repeat(rep, 1)
    assert ((M + 127) // 128) * (N // 128) <= 256 < ((A.shape[0] + 127) // 128) * (N // 128)

    def both(**kw):
        big_kw = {k: (v.repeat(rep, 1) if k in ("residual",) else v) for k, v in kw.items()}
        small = K.gemm_nt(a, b, **kw)
        big = K.gemm_nt(A, b, **big_kw)
        return small, big[:M]

    for name, kw in [("plain", {}), ("f32", {"out_f32": True}), ("bias+res", {"bias": bias, "residual": res}),
                     ("alpha-beta", {"residual": res, "alpha": 0.5, "beta": -1.5})]:
        s, bg = both(**kw)
        assert torch.equal(s, bg), name
    # epilogues with an aux tensor: separate buffers for the two calls
    aux_s = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); aux_b = torch.empty(A.shape[0], N, dtype=torch.bfloat16, device=DEV)
    for epi in (L.EPI_GELU, L.EPI_GELU_GRAD):
        s = K.gemm_nt(a, b, bias=bias, epilogue=epi, aux=aux_s)
        bg = K.gemm_nt(A, b, bias=bias, epilogue=epi, aux=aux_b)
        assert torch.equal(s, bg[:M]) and torch.equal(aux_s, aux_b[:M]), epi
        back = L.EPI_DGELU if epi == L.EPI_GELU else L.EPI_MUL_AUX
        assert torch.equal(K.gemm_nt(a, b, epilogue=back, aux=aux_s), K.gemm_nt(A, b, epilogue=back, aux=aux_b)[:M]), back
    # against fp32 math
    ref = a.float().cpu() @ b.float().cpu().t()
    got = K.gemm_nt(a, b, out_f32=True).cpu()
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-6


def test_ring_nt_batched_strided():
    g = torch.Generator().manual_seed(5)
    a = bf(torch.randn(6, 200, 256, generator=g)).to(DEV); b = bf(torch.randn(6, 128, 256, generator=g)).to(DEV)
    out = K.gemm_nt(a.view(-1, 256), b.view(-1, 256), M=200, N=128, K=256, batch=6, strideA=200 * 256, strideB=128 * 256, out_f32=True)
    ref = torch.einsum("bmk,bnk->bmn", a.float().cpu(), b.float().cpu())
    assert (out.view(6, 200, 128).cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


def _splitk_bytes(M, N, Kd):
    args = L.GemmArgs()
    args.A = args.B = args.C = 1 << 20
    args.lda, args.ldb, args.ldc = Kd, Kd, N
    args.M, args.N, args.K, args.batch, args.alpha, args.beta = M, N, Kd, 1, 1.0, 1.0
    return L.load().wft_gemm_nt_splitk_workspace_bytes(C.byref(args))


@pytest.mark.parametrize("M,N,Kd", [(1024, 512, 51968), (128, 256, 4096), (300, 384, 51968 + 64), (1024, 1280, 51968)])
def test_nt_split_k_matches_fp32_and_is_reproducible(M, N, Kd):
    assert _splitk_bytes(M, N, Kd) > 0
    g = torch.Generator().manual_seed(M + Kd)
    a = bf(torch.randn(M, Kd, generator=g)).to(DEV); b = bf(torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(DEV)
    out = K.gemm_nt(a, b)
    unsplit = K.gemm_nt(a, b, out_f32=True)  # (fp32 C: never split)
    ref = (a.double() @ b.double().t())
    assert (unsplit.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    # the split result is the bf16 rounding of an fp32 sum in another order: within one bf16 ulp (2^-7 relative) of the unsplit one
    err = (out.float() - unsplit).abs()
    assert (err <= unsplit.abs() * 2 ** -7 + 1e-6 * ref.abs().max()).all()
    assert (out.double() - ref).abs().max() <= 4e-3 * ref.abs().max()
    assert torch.equal(out, K.gemm_nt(a, b))
    # alpha is applied to the partials; a non-contiguous C (ldc > N) is honoured by the reduce
    wide = torch.full((M, N + 128), 7.0, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(a, b, out=wide[:, :N], alpha=0.5)
    assert (wide[:, N:] == 7.0).all()
    assert (wide[:, :N].float() - 0.5 * unsplit).abs().max() <= 4e-3 * ref.abs().max()


def test_nt_split_k_is_for_plain_low_occupancy_products_only():
    assert _splitk_bytes(12288, 1280, 51968) == 0   # 480 tiles: fills the chip (and is a 256 x 256 kernel problem)
    assert _splitk_bytes(1024, 512, 2048) == 0      # shallow K
    assert _splitk_bytes(1024, 512, 51968) == 8 * 1024 * 512 * 4
    assert _splitk_bytes(4096, 128, 5120) == 0      # N = 128: the rank-r adapter products stay bit-identical to their p_valid form
    g = torch.Generator().manual_seed(1)
    a = bf(torch.randn(256, 8192, generator=g)).to(DEV); b = bf(torch.randn(256, 8192, generator=g) / 90).to(DEV)
    bias = torch.randn(256, generator=g).to(DEV)
    ref = a.float().cpu() @ b.float().cpu().t() + bias.cpu()
    got = K.gemm_nt(a, b, bias=bias)   # a bias: the unsplit ring kernel
    assert (got.float().cpu() - ref).abs().max() <= 1e-2 * ref.abs().max()


def test_ring_tn_is_bit_identical_to_the_two_buffer_kernel():
    """Unsplit (short reductions: fewer than eight steps), the same output columns computed alone (6 tiles: ring) and inside a
    product of 260 tiles (two-buffer kernel): bit for bit, ragged reduction length included."""
    g = torch.Generator().manual_seed(11)
    for R in (437, 64, 130):
        a = bf(torch.randn(R, 256, generator=g)).to(DEV); b = bf(torch.randn(R, 384, generator=g)).to(DEV)
        wide = b.repeat(1, 44)[:, :128 * 130].contiguous()
        small = K.gemm_tn(a, b)
        big = K.gemm_tn(a, wide)
        assert torch.equal(small, big[:, :384]), R
        ref = a.float().cpu().t() @ b.float().cpu()
        assert (small.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


@pytest.mark.parametrize("R,P,Q", [(1024, 512, 512), (1000, 512, 2048), (12000, 1536, 512), (12000, 512, 512), (4097, 384, 640), (250, 256, 128),
                                   (1, 256, 256)])
def test_ring_tn_split_matches_fp32_and_is_reproducible(R, P, Q):
    g = torch.Generator().manual_seed(R + P)
    a = bf(torch.randn(R, P, generator=g)).to(DEV); b = bf(torch.randn(R, Q, generator=g)).to(DEV)
    ref = a.double().t() @ b.double()
    out = K.gemm_tn(a, b)
    assert (out.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert torch.equal(out, K.gemm_tn(a, b))
    base = torch.randn(P, Q, generator=g).to(DEV)
    acc = K.gemm_tn(a, b, out=base.clone(), accumulate=True, alpha=0.5)
    assert (acc.double() - (base.double() + 0.5 * ref)).abs().max() <= 2e-5 * ref.abs().max()
    # batched reduction (conv-style: several [R, *] items summed into one product)
    if R >= 64 and R % 2 == 0:
        h = R // 2
        both = K.gemm_tn(a, b, R=h, batch=2, strideA=h * P, strideB=h * Q)
        assert (both.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
