"""BASELINE configs[2] / configs[3] model-level parity: the whisper-large-v3 32/32-layer engine (B = 1, S = 128) against
the CPU oracle — loss and EVERY per-tensor gradient.

Every gradient tensor is compared with TWO oracle evaluations — the plain fp32 restatement (the parity reference proper)
and its bf16-emulation mode (oracle/whisper_oracle.py: rounds where the kernels round, fp32 accumulation) — and the bound is
conditioning-aware:

    cond(t)  = relL2(grad_emulated(t), grad_fp32(t))      how much bf16 rounding alone moves this tensor (measured, per tensor;
               smoothed: at least the median over the tensors of the same role in the other layers — a single tensor's
               value is itself a noisy estimate of its noise scale)
    relL2(gpu, emulated) <= 2e-2 + 3 cond(t)              relL2(gpu, fp32) <= 3e-2 + 3 cond(t)

Measured on this model (64 layers deep, random init): median cond = 1.8 %, but the q / k projections of the deep decoder
self-attention blocks reach 13-19 % — near-uniform causal attention makes their gradients a small difference of large
terms, so ANY two bf16 evaluations (two kernels, or kernel and emulation) differ by that much (rank-16 adapter gradients: up
to 26 %).  A mis-scaled or missing term moves a tensor by 50-100 %: it fails the bound wherever cond < 0.15-0.3, i.e. on
the same role in most layers, and it moves the MEDIANS over all tensors, which are bounded separately (full fine-tune
1.5e-2 / 2.5e-2, adapters 3e-2 / 3e-2).  The CPU oracle takes ~30-60 s per forward+backward at this size.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402
from whisper_finetune.data.gpu_frontend import GpuFrontend  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine.whisper_model import MODEL_DIMS, ModelDimensions, Whisper  # noqa: E402
from whisper_finetune.model import lora as lora_mod  # noqa: E402
from whisper_finetune.model import model_utils  # noqa: E402
from whisper_finetune.model.optimizer import get_optimizer  # noqa: E402

DEV = torch.device("cuda:0")
EMU_FLOOR, EMU_MED = 2e-2, 1.5e-2   # vs the bf16-emulating oracle (+ 2 cond per tensor)
F32_FLOOR, F32_MED = 3e-2, 2.5e-2   # vs the fp32 oracle
# No tensor may be further than this from either oracle, whatever its conditioning (1.5 x the measured maximum per
# role class).  The decoder's self-attention q / k projections and their adapters — the ill-conditioned class: 0.195 (full fine-tune) and
# 0.249 (rank-16 adapters) measured against the emulated oracle in round 6 — keep 0.35; every other tensor gets ALLOW_CAP_OTHER.
ALLOW_CAP = 0.35
ALLOW_CAP_OTHER = 0.10              # 1.5 x 0.063, the round-6 maximum over all five tests of this file (rank-16 adapters of the LoRA + Muon
#                                     configuration; full fine-tune 0.048, B = 2 0.055, whisper-small 0.020, whisper-medium 0.029)


def rel(a, b):
    # the norms on the accelerator (three comparisons of 1.5e9 values each: 1 s there, 18 s on the host); plain torch ops
    a, b = a.detach().to(DEV, torch.float32), b.detach().to(DEV, torch.float32)
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def _large_v3_params(seed=7):
    dims = O.DIMS["large-v3"]
    params = O.init_params(dims, seed=seed, device=DEV)  # drawn on the accelerator, returned as CPU tensors
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in params.items():  # non-trivial biases / LayerNorm gains so that every gradient term is exercised
        if k.endswith("bias"):
            params[k] = torch.randn(v.shape, generator=g) * 0.02
        elif "ln" in k and k.endswith("weight"):
            params[k] = 1 + torch.randn(v.shape, generator=g) * 0.05
    return dims, params


def _both_oracles(run):
    """run(emulate) -> result for emulate in (True, False), the two oracle passes side by side on two host threads (torch's CPU
    kernels release the GIL; one pass does not fill the host's cores at B = 1).  -> {True: ..., False: ...}"""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:
        futs = {e: ex.submit(run, e) for e in (True, False)}
        return {e: f.result() for e, f in futs.items()}


def _oracle_grads(dims, params, mel, y_in, y_out, emulate, lora=None, **fwd_kw):
    """loss and gradients of one oracle forward/backward; `params` values that require grad are the leaves."""
    orc = O.Oracle(dims, params, lora=lora, emulate_bf16=emulate)
    loss = O.cross_entropy(orc.forward(mel, y_in, **fwd_kw), y_out, 0.1)
    loss.backward()
    return loss.item()


def _report(errs, tag):
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(f"[{tag}] max {worst[0][1]:.4f} median {np.median(list(errs.values())):.4f} worst {worst}")
    return worst


def _check(got, emu, f32, tag, emu_med=EMU_MED, f32_med=F32_MED):
    """got / emu / f32: {name: gradient}.  Conditioning-aware per-tensor bounds (module docstring) + medians."""
    import re
    from collections import defaultdict

    raw = {n: rel(emu[n], f32[n]) for n in got}
    roles = defaultdict(list)
    for n, c in raw.items():
        roles[re.sub(r"blocks\.\d+\.", "blocks.*.", n)].append(c)
    cond = {n: max(c, float(np.median(roles[re.sub(r"blocks\.\d+\.", "blocks.*.", n)]))) for n, c in raw.items()}
    e_emu = {n: rel(got[n], emu[n]) for n in got}
    e_f32 = {n: rel(got[n], f32[n]) for n in got}
    _report(raw, tag + " cond (emulated vs fp32 oracle)")
    _report(e_emu, tag + " gpu vs emulated")
    _report(e_f32, tag + " gpu vs fp32")
    # (the allowance is capped: a sign / scale slip moves a tensor by 50-100 % and must not hide behind 3 cond on the worst-
    # conditioned q / k tensors — VERDICT r3 item 7)
    ill = re.compile(r"decoder\.blocks\.\d+\.attn\.(query|key)\.")
    cap_of = lambda n: ALLOW_CAP if ill.search(n) else ALLOW_CAP_OTHER
    others = [n for n in got if not ill.search(n)]
    if others:
        print(f"[{tag}] largest error outside the decoder's self-attention q / k: vs emulated {max(e_emu[n] for n in others):.4f}, "
              f"vs fp32 {max(e_f32[n] for n in others):.4f} (cap {ALLOW_CAP_OTHER})")
    bad = [(n, e_emu[n], e_f32[n], cond[n]) for n in got
           if e_emu[n] > min(EMU_FLOOR + 3 * cond[n], cap_of(n)) or e_f32[n] > min(F32_FLOOR + 3 * cond[n], cap_of(n))]
    assert not bad, sorted(bad, key=lambda t: -t[1])[:8]
    # per ROLE (one parameter name over all blocks): the median over the blocks is insensitive to the few badly conditioned
    # layers that set the per-tensor allowance above, so it gets the tight bound — a factor-2 slip in one role (a mis-scaled
    # term in the decoder's q / k gradients, say) moves that role's median by tens of per cent
    role_of = lambda n: re.sub(r"blocks\.\d+\.", "blocks.*.", n)
    by_role = defaultdict(lambda: ([], [], []))
    for n in got:
        r3 = by_role[role_of(n)]
        r3[0].append(e_emu[n]); r3[1].append(e_f32[n]); r3[2].append(raw[n])
    role_rows = {r: (float(np.median(a)), float(np.median(b)), float(np.median(c))) for r, (a, b, c) in by_role.items()}
    worst_roles = sorted(role_rows.items(), key=lambda kv: -(kv[1][0] - kv[1][2]))[:4]
    print(f"[{tag}] per-role medians (gpu vs emulated, gpu vs fp32, emulated vs fp32), largest excess first: {worst_roles}")
    bad_roles = [(r, v) for r, v in role_rows.items() if v[0] > EMU_FLOOR + 1.5 * v[2] or v[1] > F32_FLOOR + 1.5 * v[2]]
    assert not bad_roles, bad_roles[:6]
    assert float(np.median(list(raw.values()))) < 3e-2, "the model / inputs are too ill-conditioned for this test to mean anything"
    assert float(np.median(list(e_emu.values()))) < emu_med
    assert float(np.median(list(e_f32.values()))) < f32_med


def test_large_v3_full_finetune_step_matches_oracle():
    """configs[3] arithmetic (reference: model/model_utils.py:54-73): log-mel on the GPU, forward, label-smoothed CE,
    backward of the full 32/32 model — all 1 259 parameter tensors compared."""
    dims, params = _large_v3_params()
    audio, y_in, y_out = O.synthetic_batch(dims, 1, 128)
    y_out[0, :3] = -100
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(params)
    m.to(DEV).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    assert (mel.cpu() - mel_ref).abs().max() < 2e-3
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    del m
    torch.cuda.empty_cache()
    mel_cpu = mel.cpu()

    def run(emulate):
        p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
        loss_ref = _oracle_grads(dims, p_req, mel_cpu if emulate else mel_ref, y_in, y_out, emulate)
        return loss_ref, {n: p_req[n].grad for n in got}

    res = _both_oracles(run)
    refs = {}
    for emulate, ltol in ((True, 1e-3), (False, 2e-3)):
        loss_ref, refs[emulate] = res[emulate]
        assert abs(loss.item() - loss_ref) < ltol * loss_ref, (emulate, loss.item(), loss_ref)
    assert len(got) == len([k for k in params if k != "encoder.positional_embedding"])
    _check(got, refs[True], refs[False], "full-FT")


@pytest.mark.parametrize("name", ["small", "medium"])
def test_small_and_medium_full_finetune_step_matches_oracle(name):
    """whisper-small (d = 768, 12 heads, 12 + 12 layers: the reference's configs/DEBUG.yaml:2 `init_name: small`) and whisper-medium
    (d = 1024, 16 heads, 24 + 24 layers) through the engine — LayerNorm column templates, GEMM tile plans and head counts that no other
    model-level test touches.  2 clips, S = 32 with a ragged -100 prefix; every gradient tensor against both
    oracle evaluations with the conditioning-aware bounds of this file (a flat 8e-2 on the decoder's self-attention q / k does not hold
    from 12 layers on: 8.6 % measured on whisper-small's deepest block against the fp32 oracle, two bf16 evaluations of it differ by as
    much), plus the teacher-forced argmax of the evaluation path."""
    dims = O.DIMS[name]
    assert (dims.n_audio_state, dims.n_audio_head, dims.n_text_layer) == {"small": (768, 12, 12), "medium": (1024, 16, 24)}[name]
    params = O.init_params(dims, seed=11)
    g = torch.Generator().manual_seed(12)
    for k, v in params.items():
        if k.endswith("bias"):
            params[k] = torch.randn(v.shape, generator=g) * 0.02
        elif "ln" in k and k.endswith("weight"):
            params[k] = 1 + torch.randn(v.shape, generator=g) * 0.05
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 32)
    y_out[1, :3] = -100
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(params)
    m.to(DEV).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    mel_cpu = mel.cpu()

    def run(emulate):
        p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
        loss_ref = _oracle_grads(dims, p_req, mel_cpu if emulate else mel_ref, y_in, y_out, emulate)
        return loss_ref, {n: p_req[n].grad for n in got}

    res = _both_oracles(run)
    for emulate, ltol in ((True, 1e-3), (False, 2e-3)):
        assert abs(loss.item() - res[emulate][0]) < ltol * res[emulate][0], (emulate, loss.item(), res[emulate][0])
    assert len(got) == len([k for k in params if k != "encoder.positional_embedding"])
    _check(got, res[True][1], res[False][1], name)
    # teacher-forced argmax on the evaluation path: the oracle's tokens wherever its top-2 margin exceeds bf16 resolution
    m.eval()
    with torch.no_grad():
        logits = m(mel, y_in.to(DEV)).float().cpu()
        ref = O.Oracle(dims, params).forward(mel_ref, y_in)
    top2 = ref.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.05 * ref.abs().amax(-1)
    assert clear.any() and torch.equal(logits.argmax(-1)[clear], ref.argmax(-1)[clear])
    assert rel(logits, ref) < 2e-2


def test_large_v3_full_finetune_batch_of_two_ragged_targets_matches_fp32_oracle():
    """The same step at B = 2 with different -100 tails per row (a batch-stride slip that only shows at
    d = 1280, B > 1 was visible to the property tests only).  One fp32 oracle pass (no emulated twin: the oracle takes twice as
    long at B = 2), so the bounds are the unconditioned ones: every tensor within ALLOW_CAP, every role's median over the blocks
    within 6e-2 (3e-2 + 1.5 x the ~2 % a bf16 evaluation moves a typical tensor; 0.2 for the decoder's self-attention q / k), the
    median over all tensors within F32_MED."""
    import re
    from collections import defaultdict

    dims, params = _large_v3_params(seed=29)
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 128, seed=5)
    y_out[0, :3] = -100
    y_out[0, 100:] = -100   # ragged tails: row 0 keeps 97 targets, row 1 keeps 120
    y_out[1, 120:] = -100
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(params)
    m.to(DEV).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    del m
    torch.cuda.empty_cache()
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    loss_ref = _oracle_grads(dims, p_req, mel_ref, y_in, y_out, False)
    assert abs(loss.item() - loss_ref) < 2e-3 * loss_ref, (loss.item(), loss_ref)
    errs = {n: rel(got[n], p_req[n].grad) for n in got}
    _report(errs, "full-FT B=2 gpu vs fp32")
    ill = re.compile(r"decoder\.blocks\.\d+\.attn\.(query|key)\.")
    print(f"[full-FT B=2] largest error outside the decoder's self-attention q / k: {max(e for n, e in errs.items() if not ill.search(n)):.4f}")
    bad = {n: e for n, e in errs.items() if e > (ALLOW_CAP if ill.search(n) else ALLOW_CAP_OTHER)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]
    roles = defaultdict(list)
    for n, e in errs.items():
        roles[re.sub(r"blocks\.\d+\.", "blocks.*.", n)].append(e)
    # (the q / k projections of the decoder's causal self-attention are the ill-conditioned roles of this model: two bf16
    # evaluations of their gradients differ by 13-19 % in the deep blocks — measured by the B = 1 test's emulated pass)
    loose = ("decoder.blocks.*.attn.query", "decoder.blocks.*.attn.key")
    bad_roles = {r: float(np.median(v)) for r, v in roles.items() if float(np.median(v)) > (0.2 if r.startswith(loose) else 6e-2)}
    assert not bad_roles, bad_roles
    assert float(np.median(list(errs.values()))) < F32_MED


def test_large_v3_lora_muon_config_step_matches_oracle():
    """configs[2] = configs/config_large_v3_best_muon.yaml with model.lora: true: LoRA r16 alpha32 on all 512 Linears with a
    FIXED non-trivial dropout mask per Linear (p = 0.1), SpecAugment with the YAML's parameters (time 100, freq 43, warp 80),
    deep SpecAugment (100 / 43) and stochastic depth 0.1 with the host draws replayed into the oracle, label smoothing 0.1;
    then ONE Muon + auxiliary-Adam step (the YAML's optimizer block) against the oracle's restatement of the update."""
    dims, params = _large_v3_params(seed=17)
    t_cfg = {"stochastic_depth": 0.1}
    audio, y_in, y_out = O.synthetic_batch(dims, 1, 128, seed=99)
    r, alpha, p_drop = 16, 32, 0.1
    from whisper_finetune.model.model_utils import CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder

    m = Whisper(ModelDimensions(**vars(dims)))
    m.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head,
                                                   dims.n_audio_layer, t_cfg["stochastic_depth"])
    m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                  dims.n_text_layer, t_cfg["stochastic_depth"])
    m.load_state_dict(params)
    torch.manual_seed(17)  # lora_A's kaiming init draws from the global generator: the same adapters in every run
    lora_mod.apply_lora(m, {"rank": r, "lora_alpha": alpha, "lora_dropout": p_drop})
    g = torch.Generator().manual_seed(23)
    adapters = {}
    for name, mod in m.named_modules():
        if "parametrizations" in mod._modules:
            ad = mod.parametrizations.weight[0]
            with torch.no_grad():
                ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=g) * 0.02)
            mask = (torch.rand(1, ad.lora_A.shape[1], generator=g) >= p_drop).float() / (1.0 - p_drop)
            adapters[name] = (ad, mask)
    assert len(adapters) == 512
    m.to(DEV).train()
    for ad, mask in adapters.values():
        dev_mask = mask.to(DEV)
        ad.draw_mask = (lambda mk: (lambda training: mk))(dev_mask)
    model_utils.register_deep_spec_augment_hooks(m, time_mask_param=100, freq_mask_param=43, p=1.0)

    # SpecAugment on the device with the YAML's parameters; the drawn spans are replayed into the oracle's restatement
    fe = GpuFrontend(dims.n_mels, DEV, True, {"time_mask_param": 100, "freq_mask_param": 43, "time_warp_w": 80, "p": 1.0})
    torch.manual_seed(1234)
    sa_params, sa_ext = fe.draw(1)
    mel_plain = fe.log_mel(audio.to(DEV))
    mel = K.specaug(mel_plain, sa_params.to(DEV), sa_ext.to(DEV))
    _, wp, wd, t0, t1, f0, f1, _ = sa_params[0].tolist()
    mel_ref = O.spec_augment(O.log_mel_spectrogram(audio, dims.n_mels)[0], (wp, wd), (t0, t1), (f0, f1))[None]
    assert (mel.cpu() - mel_ref).abs().max() < 5e-3

    # replay of the model's host draws: per encoder block [skip?] then, for kept blocks but the last, the two mask spans;
    # per decoder block [skip?]  (model/model_utils.py:239,402-417; torchaudio draw order)
    torch.manual_seed(4242)
    state = torch.get_rng_state()
    enc_skips, ln_masks = [], {}
    for i in range(dims.n_audio_layer):
        s = torch.rand(1).item() < 0.1
        enc_skips.append(s)
        if not s and i < dims.n_audio_layer - 1:
            ln_masks[i] = O.draw_mask_span(100, dims.n_audio_ctx) + O.draw_mask_span(43, dims.n_audio_state)
    dec_skips = [torch.rand(1).item() < 0.1 for _ in range(dims.n_text_layer)]
    torch.set_rng_state(state)
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    named = dict(m.named_parameters())
    assert all(p.grad is None for n, p in named.items() if "lora" not in n)  # base stays frozen

    fwd_kw = dict(enc_sd_p=0.1, enc_training=True, enc_skips=enc_skips, enc_ln_masks=ln_masks,
                  dec_sd_p=0.1, dec_training=True, dec_skips=dec_skips)
    refs, skipped = {}, set()
    mel_cpu = mel.cpu()
    cfgs = {e: {n: (ad.lora_A.detach().cpu().clone().requires_grad_(True), ad.lora_B.detach().cpu().clone().requires_grad_(True),
                    ad.scaling, mask) for n, (ad, mask) in adapters.items()} for e in (True, False)}
    res = _both_oracles(lambda e: _oracle_grads(dims, params, mel_cpu if e else mel_ref, y_in, y_out, e, lora=cfgs[e], **fwd_kw))
    for emulate, ltol in ((True, 1e-3), (False, 2e-3)):
        cfg, loss_ref = cfgs[emulate], res[emulate]
        assert abs(loss.item() - loss_ref) < ltol * loss_ref, (emulate, loss.item(), loss_ref)
        refs[emulate] = {}
        for n, (A, Bm, _, _) in cfg.items():
            ad = adapters[n][0]
            if A.grad is None:  # a block dropped by stochastic depth: no gradient on either side
                assert ad.lora_A.grad is None and ad.lora_B.grad is None, n
                skipped.add(n)
                continue
            refs[emulate][n + ".lora_A"], refs[emulate][n + ".lora_B"] = A.grad, Bm.grad
    got = {}
    for n, (ad, _) in adapters.items():
        if n not in skipped:
            got[n + ".lora_A"], got[n + ".lora_B"] = ad.lora_A.grad.detach().cpu(), ad.lora_B.grad.detach().cpu()
    _check(got, refs[True], refs[False], "LoRA", emu_med=3e-2, f32_med=3e-2)
    n_skipped_blocks = sum(enc_skips) + sum(dec_skips)
    assert (len(skipped) > 0) == (n_skipped_blocks > 0)

    # ---- one optimizer step with the YAML's Muon + auxiliary-Adam block, on the ENGINE's gradients (isolates the update)
    opt_conf = {"type": "adamw", "muon": True, "8bit": False, "muon_ndim_threshold": 2,
                "muon_params": {"lr": 2e-5, "momentum": 0.95, "weight_decay": 0.01},
                "params": {"lr": 2e-5, "weight_decay": 0.01, "betas": [0.9, 0.98], "eps": 1e-6, "amsgrad": False}}
    opt = get_optimizer(m, opt_conf, is_lora_run=True)
    before = {n: p.detach().cpu().clone() for n, p in named.items() if p.requires_grad}
    names_of = {id(p): n for n, p in named.items()}

    def oracle_groups():
        groups, ref_p = [], {}
        for group in opt.param_groups:
            pairs = []
            for p in group["params"]:
                n = names_of[id(p)]
                ref_p[n] = before[n].clone()
                pairs.append((ref_p[n], None if p.grad is None else p.grad.detach().cpu().clone()))
            groups.append({**{k: v for k, v in group.items() if k != "params"}, "params": pairs})
        return groups, ref_p

    g_bf16, ref_bf16 = oracle_groups()
    g_f32, ref_f32 = oracle_groups()
    opt.step()
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:  # side by side, like the two oracle passes above
        f1 = ex.submit(O.muon_with_aux_adam_step, g_bf16, {})                          # the package's arithmetic: Newton-Schulz in bf16
        f2 = ex.submit(O.muon_with_aux_adam_step, g_f32, {}, ns_dtype=torch.float32)   # same iteration in fp32: the bf16 sensitivity
        f1.result(); f2.result()
    # ---- what the update must satisfy (the old bound 6e-2 + 2 x sensitivity reached 3.8 on rank-16 adapter
    # gradients, where two bf16 Newton-Schulz evaluations differ by up to 1.9 in relative L2 — the negated update would have
    # passed).  Per Muon matrix: SPECTRAL checks of the update against the gradient that went in (tests/_muon_spectral.py: gains
    # along the strong singular pairs in the Newton-Schulz-5 band, on the scalar iteration; update inside the gradient's row /
    # column space up to the oracle's own bf16 leak; strong-subspace projection within 5e-2 of the oracle's; <U, G> > 0), the
    # relative L2 bound 6e-2 only where the measured bf16 sensitivity is below 5e-2, and the three wrong updates (negated,
    # un-scaled, un-normalised) as negative controls that must FAIL.  Auxiliary-Adam parameters: plain relative L2 (2e-3).
    from tests._muon_spectral import negative_controls, spectral_violations

    muon_names = {names_of[id(p)] for grp in opt.param_groups if grp["use_muon"] for p in grp["params"]}
    lr_of = {names_of[id(p)]: (grp["lr"], grp["weight_decay"]) for grp in opt.param_groups for p in grp["params"]}
    errs, cond, bad, n_spec, n_l2, ctl_missed = {}, {}, [], 0, 0, []
    for n, want in ref_bf16.items():
        lr, wd = lr_of[n]
        d_got, d_want, d_f32 = named[n].detach().cpu() - before[n], want - before[n], ref_f32[n] - before[n]
        if d_want.norm() == 0:
            assert d_got.norm() == 0, n
            continue
        errs[n] = ((d_got - d_want).norm() / d_want.norm()).item()
        cond[n] = ((d_want - d_f32).norm() / d_f32.norm()).item()
        if n not in muon_names:
            if errs[n] > 2e-3:
                bad.append((n, "aux adam", errs[n]))
            continue
        if named[n].grad is None or named[n].grad.norm() == 0:  # a block dropped by stochastic depth: weight decay only
            if errs[n] > 2e-3:
                bad.append((n, "decay-only update", errs[n]))
            continue
        G = named[n].grad.detach().cpu().float()
        # the update itself: p' = p (1 - lr wd) - lr U  (first step: the momentum-mixed gradient is a multiple of G)
        shrink = before[n].double() * (1 - lr * wd)
        U_got = (shrink - named[n].detach().cpu().double()) / lr
        U_ref = (shrink - want.double()) / lr
        v = spectral_violations(U_got, G, U_ref)
        n_spec += 1
        if v:
            bad.append((n, "spectral", v))
        if cond[n] < 5e-2:
            n_l2 += 1
            if errs[n] > 6e-2:
                bad.append((n, "relative L2 of a well-conditioned update", errs[n], cond[n]))
        if n_spec <= 48 or n_spec % 16 == 0:  # negative controls on a sample of the 1 000+ matrices (each one SVD more)
            for cname, wrong in negative_controls(U_got, G).items():
                if not spectral_violations(wrong, G, U_ref):
                    ctl_missed.append((n, cname))
    _report(cond, "Muon step: bf16 vs fp32 Newton-Schulz in the oracle (sensitivity)")
    _report(errs, "Muon step: gpu vs oracle (update relative error)")
    print(f"[Muon step] spectral checks on {n_spec} matrices, relative-L2 bound applied to {n_l2} well-conditioned ones")
    assert n_spec >= 800, n_spec
    assert not bad, bad[:8]
    assert not ctl_missed, ctl_missed[:8]
