"""One-byte gelu' (WFT_EPI_GELU_GRAD8 / WFT_EPI_MUL_AUX8, include/wft.h): the MLP's forward GEMM stores the GELU
derivative as code = round(200 gelu') + 26 in gemm_nt4w_kernel's fragment order, the backward-data GEMM of mlp.2 decodes and multiplies.
Reference arithmetic: `F.gelu` between whisper.model.ResidualAttentionBlock's mlp.0 and mlp.2 (keys: scripts/convert_openai_to_hf.py:91-92)
and its autograd derivative.  Checked: the activation output is bit-identical to the bf16 pair's, every code decodes to within half a grid
step of the true derivative, the backward epilogue is bit-exact against (fp32 accumulator) x decode(code), the fused column sums, and the
effect on an MLP's gradients against fp32 math next to the bf16 pair's."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402
from whisper_finetune.engine import ops  # noqa: E402

DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


def _dgelu(x):
    x = x.double()
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5


def _code_index(M, N):
    """byte offset of element (m, n) in the fragment-ordered buffer (wft.h / gemm_nt4w.hip): int64 [M, N]"""
    m = torch.arange(M, device=DEV).view(-1, 1)
    n = torch.arange(N, device=DEV).view(1, -1)
    tiles_n = N // 256
    tm, r = m // 256, m % 256
    tn, c = n // 256, n % 256
    wave = (r // 128) * 2 + c // 128
    fx, mr = (r % 128) // 16, r % 16
    cc = c % 128
    u, q, e = cc // 32, (cc % 32) // 8, cc % 8
    up, hh = u // 2, u % 2
    lane = q * 16 + mr
    return ((tm * tiles_n + tn) * 4 + wave) * 16384 + (up * 8 + fx) * 1024 + lane * 16 + hh * 8 + e


def _decode(codes, M, N):
    return (codes[_code_index(M, N)].float() - 26.0) / 200.0


@pytest.mark.parametrize("M,N,Kd", [(8192 + 112, 1024, 768), (4500, 5120, 1280), (33000, 1536, 384)])
def test_codes_activation_and_backward_epilogue(M, N, Kd):
    g = torch.Generator(device=DEV).manual_seed(M)
    x = bf(torch.randn(M, Kd, device=DEV, generator=g))
    w = bf(torch.randn(N, Kd, device=DEV, generator=g) * (2.0 / Kd ** 0.5))
    bias = torch.randn(N, device=DEV, generator=g) * 0.5
    nb = K.gemm_nt_aux8_bytes(M, N, Kd, DEV)
    assert nb == ((M + 255) // 256) * (N // 256) * 65536, "the shape runs on gemm_nt4w_kernel"
    codes = torch.full((nb,), 255, dtype=torch.uint8, device=DEV)
    act8 = K.gemm_nt(x, w, bias=bias, epilogue=L.EPI_GELU_GRAD8, aux=codes)
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    act = K.gemm_nt(x, w, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)
    assert torch.equal(act8, act)
    pre = K.gemm_nt(x, w, bias=bias, out_f32=True)  # (the 8-wave kernel: same k order, bit-identical accumulators)
    dec = _decode(codes, M, N)
    true = _dgelu(pre).float()
    # the kernel's own gelu' (a polynomial erfc: the bf16 pair stores its bf16 rounding) is within 1.5e-3 of the exact derivative
    assert (dec - true).abs().max().item() <= 0.0025 + 2e-3
    assert (dec - aux.float()).abs().max().item() <= 0.0025 + 2 ** -8  # half a grid step + the bf16 rounding of the other form
    assert codes[_code_index(M, N)].min().item() >= 0 and dec.min().item() >= -0.13 and dec.max().item() <= 1.145
    # saturated units decode to exactly 0 and 1
    assert (dec[pre > 8] == 1.0).all() and (dec[pre < -8] == 0.0).all()

    # backward-data epilogue: C = bf16(acc * decode(code)), bit for bit; column sums of the values written
    Kb = 256 if Kd != 1280 else 1280
    dy = bf(torch.randn(M, Kb, device=DEV, generator=g))
    wt = bf(torch.randn(N, Kb, device=DEV, generator=g) * 0.1)
    cs = torch.full((N,), float("nan"), device=DEV)
    assert K.gemm_nt_aux8_bytes(M, N, Kb, DEV, epilogue=L.EPI_MUL_AUX8, colsum=True) == nb
    got = K.gemm_nt(dy, wt, epilogue=L.EPI_MUL_AUX8, aux=codes, colsum=cs)
    acc = K.gemm_nt(dy, wt, out_f32=True)
    gq = (codes[_code_index(M, N)].double() * float(torch.tensor(0.005, dtype=torch.float32)) + float(torch.tensor(-0.13, dtype=torch.float32))).float()
    want = bf((acc.double() * gq.double()).float())
    assert torch.equal(got, want)
    ref_cs = (acc.double() * gq.double()).sum(0).float()  # (the fused sums add the fp32 values BEFORE their bf16 rounding, like WFT_EPI_MUL_AUX)
    assert (cs - ref_cs).abs().max().item() <= 1e-4 * ref_cs.abs().max().item() + 1e-3
    got2 = K.gemm_nt(dy, wt, epilogue=L.EPI_MUL_AUX8, aux=codes)  # (the instantiation without column sums)
    assert torch.equal(got2, got)


def test_small_problems_are_not_served_in_the_byte_form():
    assert K.gemm_nt_aux8_bytes(3000, 1536, 384, DEV) == 0          # whisper-tiny at 2 clips: the 128-tile kernel
    assert K.gemm_nt_aux8_bytes(48000, 5120, 1280, DEV) > 0
    old = K.set_variant("nt", 1)
    try:
        assert K.gemm_nt_aux8_bytes(48000, 5120, 1280, DEV) == 0    # forced 8-wave kernel: bf16 pair
    finally:
        K.set_variant("nt", old)
    x = bf(torch.randn(3000, 384, device=DEV)); w = bf(torch.randn(1536, 384, device=DEV))
    with pytest.raises(L.WftError):
        K.gemm_nt(x, w, epilogue=L.EPI_GELU_GRAD8, aux=torch.empty(1 << 20, dtype=torch.uint8, device=DEV))


def _mlp_grads(aux8: bool, x, w1, b1, w2, b2, dy):
    from whisper_finetune.engine.whisper_model import MLP, Linear

    old = ops._GELU_AUX8
    ops._GELU_AUX8 = aux8
    try:
        d, h = w1.shape[1], w1.shape[0]
        mlp = MLP(Linear(d, h), torch.nn.GELU(), Linear(h, d)).to(DEV)
        with torch.no_grad():
            mlp[0].weight.copy_(w1); mlp[0].bias.copy_(b1); mlp[2].weight.copy_(w2); mlp[2].bias.copy_(b2)
        xx = x.clone().requires_grad_(True)
        seen = []
        real = K.gemm_nt

        def spy(*a, **kw):
            seen.append(kw.get("epilogue", L.EPI_NONE))
            return real(*a, **kw)

        K.gemm_nt = spy
        try:
            y = mlp(xx)
            y.backward(dy)
        finally:
            K.gemm_nt = real
        return y.detach(), xx.grad, {n: p.grad for n, p in mlp.named_parameters()}, seen
    finally:
        ops._GELU_AUX8 = old


def test_mlp_gradients_with_one_byte_derivative_against_fp32_math():
    """d = 1280, 4d = 5120, 16 500 rows (the large-v3 MLP at 11 clips): y, dx, dW1, db1, dW2, db2 of the engine with the byte form and
    with the bf16 pair, both against fp32 autograd of the same bf16-rounded operands.  The byte form may not be further from fp32 math
    than the bf16 pair by more than 2e-3 relative L2 on any tensor, and must itself stay within the engine's per-tensor bound (3e-2)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    M, d, h = 16500, 1280, 5120
    x = bf(torch.randn(M, d, device=DEV, generator=g))
    w1 = torch.randn(h, d, device=DEV, generator=g) * d ** -0.5
    b1 = torch.randn(h, device=DEV, generator=g) * 0.3
    w2 = torch.randn(d, h, device=DEV, generator=g) * h ** -0.5
    b2 = torch.randn(d, device=DEV, generator=g) * 0.1
    dy = bf(torch.randn(M, d, device=DEV, generator=g))
    xr = x.float().requires_grad_(True)
    p = [t.clone().requires_grad_(True) for t in (bf(w1).float(), b1, bf(w2).float(), b2)]
    yr = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr, p[0], p[1])), p[2], p[3])
    yr.backward(dy.float())
    ref = {"y": yr.detach(), "dx": xr.grad, "0.weight": p[0].grad, "0.bias": p[1].grad, "2.weight": p[2].grad, "2.bias": p[3].grad}
    errs = {}
    for aux8 in (True, False):
        y, dx, grads, seen = _mlp_grads(aux8, x, w1, b1, w2, b2, dy)
        assert (L.EPI_GELU_GRAD8 in seen and L.EPI_MUL_AUX8 in seen) == aux8 and ((L.EPI_GELU_GRAD in seen) != aux8)
        got = {"y": y, "dx": dx, **grads}
        errs[aux8] = {n: ((got[n].float() - ref[n]).norm() / ref[n].norm()).item() for n in ref}
    print("relative L2 vs fp32 math, one-byte gelu':", errs[True], "| bf16 gelu':", errs[False])
    assert errs[True]["y"] == errs[False]["y"]  # the forward values do not depend on the form
    for n in ref:
        assert errs[True][n] < 3e-2 and errs[True][n] <= errs[False][n] + 2e-3, (n, errs[True][n], errs[False][n])

