"""`wft_attn_args.q_prescaled` and the forward-shadow scale that feeds it: the softmax scale * log2(e) is folded
into the q rows of the fused QKV projection's FORWARD bf16 shadow (one rounding of the scaled value, what `q * scale` costs in
whisper.model.MultiHeadAttention.qkv_attention — reached from model/model_utils.py:283-285,320-322), and no attention kernel multiplies
its scores.  Checked here: the shadow kernels (per-weight, LoRA merge, batched refresh, stacked bias), the kernels against fp32 math on
exactly the operands they see, bit-identity of the one-wave-per-SIMD kernels to their 8-wave twins with the flag set, and the model-level
forward / backward against the flag switched off."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import ops  # noqa: E402
from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper  # noqa: E402

DEV = "cuda:0"
ALPHA = ops.QK_ALPHA
SCALE = 0.125


def bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(autouse=True)
def _restore_variants():
    old = dict(K.VARIANT)
    yield
    K.VARIANT.update(old)


def test_alpha_is_scale_times_log2e():
    assert ALPHA == pytest.approx(SCALE * math.log2(math.e), rel=1e-7)


def test_forward_shadow_is_one_rounding_of_the_scaled_weight():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(384, 256, generator=g).to(DEV)
    d, dt = K.weight_shadow(w, 384, 256, True, fwd_scale=ALPHA)
    assert torch.equal(d, bf(w * ALPHA)) and torch.equal(dt, bf(w).t())
    d1, dt1 = K.weight_shadow(w, 384, 256, True)
    assert torch.equal(d1, bf(w)) and torch.equal(dt1, dt)
    # ragged (element-wise path of the kernel)
    w2 = torch.randn(301, 250, generator=g).to(DEV)
    d2, dt2 = K.weight_shadow(w2, 384, 256, True, fwd_scale=ALPHA)
    ref = torch.zeros(384, 256, device=DEV); ref[:301, :250] = w2
    assert torch.equal(d2, bf(ref * ALPHA)) and torch.equal(dt2, bf(ref).t())
    # LoRA merge: W + s B (A * mask), scaled only on the straight image
    B, A = torch.randn(384, 8, generator=g).to(DEV) * 0.1, torch.randn(8, 256, generator=g).to(DEV) * 0.1
    mask = (torch.rand(256, generator=g) > 0.3).float().to(DEV) / 0.7
    out, out_t = torch.empty(384, 256, dtype=torch.bfloat16, device=DEV), torch.empty(256, 384, dtype=torch.bfloat16, device=DEV)
    K.lora_merge(w, B, A, mask, 2.0, out=out, out_t=out_t, fwd_scale=ALPHA)
    o1, o1t = torch.empty_like(out), torch.empty_like(out_t)
    K.lora_merge(w, B, A, mask, 2.0, out=o1, out_t=o1t)
    assert torch.equal(out_t, o1t)
    eff = w.double() + 2.0 * (B.double() @ (A.double() * mask.double()))
    assert (out.float() - (eff * ALPHA).float()).abs().max().item() <= 2 ** -8 * (eff * ALPHA).abs().max().item()
    assert not torch.equal(out, o1)


def _operands(B, H, Tq, Tk, seed, qkv_layout=True):
    g = torch.Generator(device=DEV).manual_seed(seed)
    d = H * 64
    q0 = torch.randn(B, Tq, d, device=DEV, generator=g) * 2
    if qkv_layout and Tq == Tk:
        buf = torch.empty(B, Tq, 3 * d, dtype=torch.bfloat16, device=DEV)
        buf[..., d:] = bf(torch.randn(B, Tq, 2 * d, device=DEV, generator=g) * 2)
        k, v = buf[..., d:2 * d], buf[..., 2 * d:]
        q_pre, q_plain = buf[..., :d], bf(q0)
        q_pre.copy_(bf(q0 * ALPHA))
    else:
        kv = bf(torch.randn(B, Tk, 2 * d, device=DEV, generator=g) * 2)
        k, v = kv[..., :d], kv[..., d:]
        q_pre, q_plain = bf(q0 * ALPHA), bf(q0)
    do = bf(torch.randn(B, Tq, d, device=DEV, generator=g))
    return q_pre, q_plain, k, v, do


def _fp32_ref(q_eff, k, v, do, H, causal):
    """softmax(scale q k^T) v per head in fp32 on the operands the kernel sees (q_eff = q_prescaled / alpha) + autograd."""
    B, Tq, d = q_eff.shape
    Tk = k.shape[1]
    qh = q_eff.float().view(B, Tq, H, 64).transpose(1, 2).detach().requires_grad_(True)
    kh = k.float().reshape(B, Tk, H, 64).transpose(1, 2).detach().requires_grad_(True)
    vh = v.float().reshape(B, Tk, H, 64).transpose(1, 2).detach().requires_grad_(True)
    s = (qh @ kh.transpose(-1, -2)) * SCALE
    if causal:
        s = s + torch.full((Tq, Tk), float("-inf"), device=DEV).triu_(1)
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ vh
    o.backward(do.float().view(B, Tq, H, 64).transpose(1, 2))
    back = lambda t: t.transpose(1, 2).reshape(B, -1, H * 64)
    return back(o), lse, back(qh.grad), back(kh.grad), back(vh.grad)


@pytest.mark.parametrize("B,H,Tq,Tk,causal", [(2, 6, 1500, 1500, False), (3, 8, 77, 77, True), (2, 4, 448, 448, True),
                                               (2, 6, 130, 1500, False), (1, 20, 640, 700, False)])
def test_prescaled_kernels_match_fp32_math(B, H, Tq, Tk, causal):
    q_pre, q_plain, k, v, do = _operands(B, H, Tq, Tk, seed=Tq + Tk)
    o, lse = K.attn_fwd(q_pre, k, v, H, causal, SCALE, q_prescaled=True)
    dq, dk, dv = K.attn_bwd(q_pre, k, v, o, lse, do, H, causal, SCALE, q_prescaled=True)
    ro, rlse, rdq, rdk, rdv = _fp32_ref(q_pre.float() / ALPHA, k, v, do, H, causal)
    assert (lse - rlse).abs().max().item() < 2e-3
    for got, want, nm, tol in ((o, ro, "o", 1e-2), (dq, rdq, "dq", 2e-2), (dk, rdk, "dk", 2e-2), (dv, rdv, "dv", 2e-2)):
        e = ((got.float() - want).norm() / want.norm()).item()
        assert e < tol, (nm, e)
    # ... and agree with the un-prescaled path on the plainly rounded q to bf16 resolution of q (two roundings of the same number)
    o2, lse2 = K.attn_fwd(q_plain, k, v, H, causal, SCALE)
    assert ((o.float() - o2.float()).norm() / o2.float().norm()).item() < 2e-2


@pytest.mark.parametrize("B,H,T", [(2, 6, 1500), (1, 8, 1100)])
def test_prescaled_4wave_kernels_are_bit_identical_to_the_8wave_twins(B, H, T):
    """Same arithmetic with the c-multiplies assembled out (one-wave-per-SIMD kernels, pipelined forward) or multiplying by exactly
    1.0 (the 8-wave kernels, which take the flag at run time)."""
    q_pre, _, k, v, do = _operands(B, H, T, T, seed=T)
    outs = []
    for var in (0, 1):
        for w in ("fwd", "dq", "dkdv"):
            K.set_variant(w, var)
        o, lse = K.attn_fwd(q_pre, k, v, H, False, SCALE, q_prescaled=True)
        cs = (torch.full((H * 64,), float("nan"), device=DEV), torch.full((H * 64,), float("nan"), device=DEV))
        dq, dk, dv = K.attn_bwd(q_pre, k, v, o, lse, do, H, False, SCALE, colsums=cs, q_prescaled=True)
        outs.append((o, lse, dq, dk, dv, cs[0], cs[1]))
    names = ("o", "lse", "dq", "dk", "dv", "cs_q", "cs_v")
    for nm, a, b in zip(names, *outs):
        if nm in ("dk", "dv", "cs_v"):  # the 4-wave dK/dV kernel sums queries in a different order (as without the flag: wft.h)
            assert ((a.float() - b.float()).norm() / b.float().norm()).item() < 1e-2, nm
        else:
            assert torch.equal(a, b), nm


def _model(name, prescale, seed=3):
    old = ops.QK_PRESCALE
    ops.QK_PRESCALE = prescale
    try:
        dims = O.DIMS[name]
        m = Whisper(ModelDimensions(**vars(dims)))
    finally:
        ops.QK_PRESCALE = old
    m.load_state_dict(O.init_params(dims, seed=seed))
    return m.to(DEV).train(), dims


def test_model_with_prescaled_q_matches_the_unscaled_path():
    """whisper-tiny, 2 clips: loss and every gradient of the prescaled engine against the engine with WFT_QK_PRESCALE=0 (two bf16
    evaluations of the same function: 3e-2 per tensor, 8e-2 for the ill-conditioned decoder q / k, as tests/test_model_gpu.py), and the
    shadows carry the factor where they should."""
    m1, dims = _model("tiny", True)
    m0, _ = _model("tiny", False)
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 24)
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    losses = []
    for m in (m1, m0):
        loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
        loss.backward()
        losses.append(loss.item())
    assert abs(losses[0] - losses[1]) < 5e-4 * losses[1]
    import re

    ill = re.compile(r"decoder\.blocks\.\d+\.attn\.(query|key)\.")
    p0 = dict(m0.named_parameters())
    for n, p in m1.named_parameters():
        if p.grad is None:
            continue
        e = ((p.grad - p0[n].grad).norm() / (p0[n].grad.norm() + 1e-20)).item()
        assert e < (8e-2 if ill.search(n) else 3e-2), (n, e)
    for blk in list(m1.encoder.blocks) + list(m1.decoder.blocks):
        g, d = blk.attn._qkv_group, blk.attn.query.weight.shape[0]
        assert g.fwd_scales == (ALPHA, 1.0, 1.0)
        assert torch.equal(g.W[:d], bf(blk.attn.query.weight.detach() * ALPHA))
        assert torch.equal(g.W[d:2 * d], bf(blk.attn.key.weight.detach())) and torch.equal(g.W[2 * d:3 * d], bf(blk.attn.value.weight.detach()))
        assert torch.equal(g.WT[:, :d], bf(blk.attn.query.weight.detach()).t())  # backward-data shadow: unscaled
        assert torch.equal(g.bias[:d], blk.attn.query.bias.detach() * ALPHA) and torch.equal(g.bias[2 * d:3 * d], blk.attn.value.bias.detach())
    for blk in m0.encoder.blocks:
        assert blk.attn._qkv_group.fwd_scales is None


def test_lora_on_q_with_prescaled_shadow_matches_the_unscaled_path():
    """LoRA r = 8 on every Linear (dropout off): the merged forward shadow W + s B A carries the factor on the q rows, the adapter
    gradients come from the unscaled dq: same loss and adapter gradients as with the flag off (bf16 tolerances of test_model_gpu)."""
    from whisper_finetune.model import lora

    res = []
    for pre in (True, False):
        m, dims = _model("tiny", pre, seed=5)
        torch.manual_seed(0)  # (lora_A: kaiming-uniform from the default generator)
        lora.apply_lora(m, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.0})
        for n, p in m.named_parameters():
            if "lora_B" in n:
                with torch.no_grad():
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(n))).to(DEV) * 0.02)
        audio, y_in, y_out = O.synthetic_batch(dims, 2, 16)
        mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
        for step in range(2):  # the second forward runs through the batched refresh plan
            m.zero_grad(set_to_none=True)
            loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.0)
            loss.backward()
        res.append((loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
    (l1, g1), (l0, g0) = res
    assert abs(l1 - l0) < 5e-4 * l0
    import re

    ill = re.compile(r"decoder\.blocks\.\d+\.attn\.(query|key)\.")
    assert g1.keys() == g0.keys() and len(g1) > 50
    for n in g1:
        e = ((g1[n] - g0[n]).norm() / (g0[n].norm() + 1e-20)).item()
        assert e < (0.12 if ill.search(n) else 5e-2), (n, e)
