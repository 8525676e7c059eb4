"""Whole-path GPU parity: the libwft engine (through whisper_finetune's reference-shaped interface) against the
golden fixture (HF transformers), the CPU oracle, and the reference's training-time extras.
bf16 tolerance (stated per check): activations are rounded to bf16 between kernels (8 mantissa bits), so
logits agree to ~1e-2 relative L2, the mean loss to 5e-4 relative, gradients per tensor to 3e-2 relative L2 (8e-2 for the
ill-conditioned q / k projections of the decoder's causal self-attention; `assert_grad_errors`)."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402
from tests.golden.gen_golden import ARCH_DIMS, arch_inputs, arch_params  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine.whisper_model import MODEL_DIMS, ModelDimensions, Whisper  # noqa: E402
from whisper_finetune.model import lora as lora_mod  # noqa: E402
from whisper_finetune.model import model_utils  # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


_ILL = re.compile(r"decoder\.blocks\.\d+\.attn\.(query|key)\.")


def assert_lora_grad_errors(errs):
    """Rank-r adapter gradients (two extra bf16 roundings: du = dy sB and u = x (sA*m)^T are bf16 GEMM outputs): every adapter
    but those of the decoder's causal self-attention q / k within 5e-2, those within 0.12, median 2e-2."""
    worst_ill = sorted(((n, e) for n, e in errs.items() if _ILL.search(n)), key=lambda kv: -kv[1])[:3]
    worst = sorted(((n, e) for n, e in errs.items() if not _ILL.search(n)), key=lambda kv: -kv[1])[:3]
    print("LoRA grad errors: worst", worst, "worst ill-conditioned", worst_ill, "median", float(np.median(list(errs.values()))))
    assert worst[0][1] < 5e-2, worst
    assert not worst_ill or worst_ill[0][1] < 0.12, worst_ill
    assert float(np.median(list(errs.values()))) < 2e-2


def assert_grad_errors(errs, tag=""):
    """Per-tensor relative-L2 gradient errors of a bf16 engine step against the oracle (fp32 or bf16-emulating).
    Calibrated on the GPU (tools/dev/emu_tol.py, tiny and base): every tensor but the q / k projections of the decoder's causal
    self-attention sits at 0.9-1.6 % (bound 3e-2 — a mis-scaled term moves a tensor by tens of per cent); those q / k tensors
    are small differences of large terms under near-uniform attention and reach 3.5-6.8 % for ANY two bf16 evaluations
    (bound 8e-2); median 0.5-0.9 % (bound 1.5e-2)."""
    worst_ill = sorted(((n, e) for n, e in errs.items() if _ILL.search(n)), key=lambda kv: -kv[1])[:3]
    worst = sorted(((n, e) for n, e in errs.items() if not _ILL.search(n)), key=lambda kv: -kv[1])[:3]
    assert worst[0][1] < 3e-2, (tag, worst)
    assert not worst_ill or worst_ill[0][1] < 8e-2, (tag, worst_ill)
    assert float(np.median(list(errs.values()))) < 1.5e-2, tag


def _engine(dims: O.ModelDimensions, params):
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(params)
    return m.to(DEV)


def test_engine_matches_hf_golden(golden_arch):
    dims = ARCH_DIMS
    m = _engine(dims, arch_params(dims, 3)).train()
    mel, y_in, y_out = arch_inputs(dims, 11)
    loss = m(mel.to(DEV), y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    ref_loss = float(golden_arch["loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * ref_loss, (loss.item(), ref_loss)
    m.eval()
    with torch.no_grad():
        logits = m(mel.to(DEV), y_in.to(DEV))
    assert logits.dtype == torch.float32 and logits.shape == (2, 12, dims.n_vocab)
    assert rel(logits, torch.from_numpy(golden_arch["logits"])) < 2e-2
    named = dict(m.named_parameters())
    for n, ref in zip([str(x) for x in golden_arch["grad_norm_names"]], golden_arch["grad_norms"]):
        got = named[n].grad.norm().item()
        assert abs(got - ref) < 6e-2 * ref + 1e-6, (n, got, ref)
    for key in golden_arch.files:
        if key.startswith("grad::"):
            assert rel(named[key[6:]].grad, torch.from_numpy(golden_arch[key])) < 6e-2, key


def _tiny_case(B=2, S=24, seed=0):
    dims = O.DIMS["tiny"]
    params = O.init_params(dims, seed=seed)
    g = torch.Generator().manual_seed(5)
    for k, v in params.items():
        if k.endswith("bias"):
            params[k] = torch.randn(v.shape, generator=g) * 0.02
        elif "ln" in k and k.endswith("weight"):
            params[k] = 1 + torch.randn(v.shape, generator=g) * 0.05
    audio, y_in, y_out = O.synthetic_batch(dims, B, S)
    y_out[0, :3] = -100
    return dims, params, audio, y_in, y_out


def test_tiny_training_step_matches_oracle():
    """configs[0]-shaped case (whisper-tiny, 2 synthetic 30 s clips): log-mel on the GPU, forward, fused
    label-smoothed CE, backward — against the fp32 CPU oracle."""
    dims, params, audio, y_in, y_out = _tiny_case()
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    loss_ref = O.cross_entropy(O.Oracle(dims, p_req).forward(mel_ref, y_in), y_out, 0.1)
    loss_ref.backward()
    m = _engine(dims, params).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    assert (mel.cpu() - mel_ref).abs().max() < 2e-3
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * loss_ref.item()
    assert_grad_errors({n: rel(p.grad, p_req[n].grad) for n, p in m.named_parameters()}, "vs fp32 oracle")
    # ... and against the oracle's bf16-emulation mode (rounds where the kernels round)
    p_emu = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    loss_emu = O.cross_entropy(O.Oracle(dims, p_emu, emulate_bf16=True).forward(mel.cpu(), y_in), y_out, 0.1)
    loss_emu.backward()
    assert abs(loss.item() - loss_emu.item()) < 5e-4 * loss_emu.item()
    assert_grad_errors({n: rel(p.grad, p_emu[n].grad) for n, p in m.named_parameters()}, "vs bf16-emulating oracle")
    # fused loss == the reference's two-step form on the same logits
    m.eval()
    with torch.no_grad():
        logits = m(mel, y_in.to(DEV))
        two_step = torch.nn.functional.cross_entropy(logits.transpose(1, 2), y_out.to(DEV), label_smoothing=0.1)
    assert abs(two_step.item() - loss.item()) < 1e-3 * loss.item()


def test_teacher_forced_argmax_is_bit_exact_on_the_same_logits():
    """eval/evaluator.py:70-73: argmax over the padded batch.  The kernel's argmax equals torch's argmax of the
    logits the engine returns (ids bit-exact); against the fp32 oracle only near-ties may differ."""
    dims, params, audio, y_in, y_out = _tiny_case(S=16)
    m = _engine(dims, params).eval()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    with torch.no_grad():
        h = m.decoder.hidden(y_in.to(DEV), m.encoder(mel))
        padded = m.decoder.padded_logits(h)
        _, _, _, am = K.ce_fwd(padded, y_out.reshape(-1).to(DEV), dims.n_vocab, 0.0, want_argmax=True)
        logits = m(mel, y_in.to(DEV))
    assert torch.equal(am.view(2, 16), logits.argmax(-1))
    ref = O.Oracle(dims, params).forward(O.log_mel_spectrogram(audio, dims.n_mels), y_in)
    lr = logits.cpu()
    top2 = ref.topk(2, -1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 0.05  # margin larger than the bf16 logit error
    assert torch.equal(lr.argmax(-1)[decisive], ref.argmax(-1)[decisive])


def test_lora_low_rank_path_matches_oracle_parametrization():
    dims, params, audio, y_in, y_out = _tiny_case()
    m = Whisper(MODEL_DIMS["tiny"]); m.load_state_dict(params)
    torch.manual_seed(9)  # lora_A's kaiming init draws from the global generator: the same adapters in every run
    lora_mod.apply_lora(m, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.0})
    gl = torch.Generator().manual_seed(9)
    cfg = {}
    for n, mod in m.named_modules():
        if "parametrizations" in mod._modules:
            ad = mod.parametrizations.weight[0]
            with torch.no_grad():
                ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=gl) * 0.05)
            cfg[n] = (ad.lora_A.detach().clone().requires_grad_(True), ad.lora_B.detach().clone().requires_grad_(True), ad.scaling, None)
    m.to(DEV).train()
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    loss_ref = O.cross_entropy(O.Oracle(dims, params, lora=cfg).forward(mel, y_in), y_out, 0.1)
    loss_ref.backward()
    loss = m(mel.to(DEV), y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * loss_ref.item()
    mods = dict(m.named_modules())
    errs = {}
    for n, (A, Bm, _, _) in cfg.items():
        ad = mods[n].parametrizations.weight[0]
        errs[n + ".lora_A"], errs[n + ".lora_B"] = rel(ad.lora_A.grad, A.grad), rel(ad.lora_B.grad, Bm.grad)
    assert_lora_grad_errors(errs)
    assert all(p.grad is None for n, p in m.named_parameters() if "lora" not in n)  # base stays frozen


def test_adapted_forward_and_backward_launch_no_torch_matmul():
    """An adapted Linear's `.weight` PROPERTY materialises W + s B A with torch ops (model/lora.py forward: merge / inspection only).
    The training path must never touch it: round 6 found `fc2.weight.shape[0]` in the MLP doing exactly that once per block and
    forward (57 hipBLASLt GEMMs per configs[2] step in `profiles/r06_a_lora_muon_kernel_stats.csv`)."""
    from torch.overrides import TorchFunctionMode

    dims, params, audio, y_in, y_out = _tiny_case()
    m = Whisper(MODEL_DIMS["tiny"]); m.load_state_dict(params)
    lora_mod.apply_lora(m, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.1})
    m.to(DEV).train()
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    seen = []

    class Spy(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            if getattr(func, "__name__", "") in ("matmul", "__matmul__", "__rmatmul__", "mm", "bmm", "addmm", "baddbmm", "linear", "einsum"):
                seen.append(func.__name__)
            return func(*args, **(kwargs or {}))

    with Spy():
        loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
        loss.backward()
    assert not seen, seen


def test_lora_dropout_mask_is_per_input_column():
    """minLoRA drops whole input columns of A (mask [1, in] shared by the batch): with a fixed mask the engine
    equals the oracle's W + s*B@(A*mask)."""
    from whisper_finetune.engine import ops
    from whisper_finetune.engine.whisper_model import Linear
    g = torch.Generator().manual_seed(0)
    lin = Linear(256, 384)
    lora_mod.add_lora_to_linear(lin, 4, 8, 0.5)
    lin.to(DEV)
    ad = lin.parametrizations.weight[0]
    with torch.no_grad():
        ad.lora_B.normal_(generator=None)
    mask = (torch.rand(1, 256, generator=g) > 0.5).float().to(DEV) / 0.5
    ad.draw_mask = lambda training: mask
    x = torch.randn(10, 256, generator=g).to(DEV)
    y = lin(x)
    W = O.lora_effective_weight(lin.parametrizations.weight.original.detach().cpu(), ad.lora_A.detach().cpu(), ad.lora_B.detach().cpu(),
                                ad.scaling, mask.cpu())
    ref = x.cpu().to(torch.bfloat16).float() @ W.T + lin.bias.detach().cpu()
    assert rel(y, ref) < 1e-2


def test_stochastic_depth_and_deep_specaug_semantics():
    """S1/A4: CheckpointedStochastic encoder/decoder + deep-SpecAugment hooks give the oracle's result when the
    same host draws are replayed (skip decisions, mask spans)."""
    dims, params, audio, y_in, y_out = _tiny_case()
    from whisper_finetune.model.model_utils import CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder
    m = Whisper(MODEL_DIMS["tiny"])
    m.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer, 0.3)
    m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer, 0.3)
    m.load_state_dict(params)
    m.to(DEV).train()
    model_utils.register_deep_spec_augment_hooks(m, time_mask_param=100, freq_mask_param=27, p=1.0)
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    # replay the host RNG to know what the engine will draw: per encoder block [skip?] then (if kept and not the last
    # block) 4 rands for the masks; then per decoder block [skip?]
    torch.manual_seed(77)
    state = torch.get_rng_state()
    enc_skips, masks = [], {}
    for i in range(dims.n_audio_layer):
        s = torch.rand(1).item() < 0.3
        enc_skips.append(s)
        if not s and i < dims.n_audio_layer - 1:
            t0, t1 = O.draw_mask_span(100, dims.n_audio_ctx)
            c0, c1 = O.draw_mask_span(27, dims.n_audio_state)
            masks[i] = (t0, t1, c0, c1)
    dec_skips = [torch.rand(1).item() < 0.3 for _ in range(dims.n_text_layer)]
    ref_logits = O.Oracle(dims, params).forward(mel, y_in, enc_sd_p=0.3, enc_training=True, enc_skips=enc_skips, enc_ln_masks=masks,
                                                dec_sd_p=0.3, dec_training=True, dec_skips=dec_skips)
    ref_loss = O.cross_entropy(ref_logits, y_out, 0.0)
    torch.set_rng_state(state)
    loss = m(mel.to(DEV), y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.0)
    assert abs(loss.item() - ref_loss.item()) < 3e-3 * ref_loss.item(), (loss.item(), ref_loss.item(), enc_skips, dec_skips)
    loss.backward()  # checkpoint recompute replays the same draws
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_train_step_loss_sequence_decreases_and_matches_first_step():
    """T: the reference-shaped train_step (accumulation 2, clip, AdamW) over the engine; first-step loss equals the
    oracle's, and a few steps on a fixed batch reduce the loss."""
    dims, params, audio, y_in, y_out = _tiny_case(B=2, S=12)
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    m = _engine(dims, params)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 2, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1}

    def batches():
        while True:
            yield mel, y_in, y_out

    first = model_utils.train_step(m, batches(), opt, sched, t_cfg)
    ref = O.train_step_loss(O.Oracle(dims, params), [(mel, y_in, y_out)] * 2, 2, 0.1).item()
    assert abs(first - ref) < 2e-3 * ref
    last = first
    for _ in range(4):
        last = model_utils.train_step(m, batches(), opt, sched, t_cfg)
    assert last < first - 0.05


def test_deferred_loss_readback_returns_the_same_floats():
    """train_step reads the micro-batch losses back through pinned slots behind an event AFTER it has enqueued clip + optimizer
    (the reference's `loss.item()` at model/model_utils.py:73 sits in front of them): same returned values, same parameters, bit for
    bit, as the synchronous read-back (`wft_defer_loss_readback: False`), with accumulation 3."""
    from whisper_finetune.model.optimizer import WftAdamW

    dims, params, audio, y_in, y_out = _tiny_case(B=2, S=12)
    mels = [K.logmel((audio * (1 + 0.1 * i)).to(DEV), O.mel_filters(dims.n_mels).to(DEV)) for i in range(3)]

    def run(defer):
        m = _engine(dims, params)
        opt = WftAdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
        t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 3, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1,
                 "wft_defer_loss_readback": defer}

        def batches():
            i = 0
            while True:
                yield mels[i % 3], y_in.to(DEV), y_out.to(DEV)
                i += 1

        it = batches()
        return [model_utils.train_step(m, it, opt, sched, t_cfg) for _ in range(3)], [p.detach().clone() for p in m.parameters()]

    l1, p1 = run(True)
    l0, p0 = run(False)
    assert l1 == l0 and all(isinstance(v, float) for v in l1)
    assert all(torch.equal(a, b) for a, b in zip(p1, p0))


def test_loss_curve_parity_with_the_cpu_oracle_over_optimizer_steps():
    """configs[0]-shaped loss-curve parity: 4 optimizer steps of train_step (accumulation 2, label smoothing, clip 1.0,
    AdamW) on the engine with the libwft optimizer vs the SAME steps on the fp32 CPU oracle with torch.optim.AdamW +
    clip_grad_norm_.  bf16 compute vs fp32: every loss of the curve within 3e-3 relative, and the parameters after the
    last step within 2e-3 relative L2 of the oracle's (updates of ~lr per element, so this bounds the accumulated drift)."""
    from whisper_finetune.model.optimizer import WftAdamW

    dims, params, audio, y_in, y_out = _tiny_case(B=2, S=12)
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    kw = dict(lr=3e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    # --- oracle side
    names = [k for k in params if k != "encoder.positional_embedding"]
    ref_p = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in params.items()}
    ref_opt = torch.optim.AdamW([ref_p[k] for k in names], **kw)
    ref_curve = []
    for _ in range(4):
        ref_opt.zero_grad(set_to_none=True)
        loss = O.train_step_loss(O.Oracle(dims, ref_p), [(mel, y_in, y_out)] * 2, 2, 0.1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([ref_p[k] for k in names], 1.0)
        ref_opt.step()
        ref_curve.append(loss.item())
    # --- engine side
    m = _engine(dims, params)
    opt = WftAdamW(m.parameters(), **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 2, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1}

    def batches():
        while True:
            yield mel, y_in, y_out

    curve = [model_utils.train_step(m, batches(), opt, sched, t_cfg) for _ in range(4)]
    for got, want in zip(curve, ref_curve):
        assert abs(got - want) < 3e-3 * want, (curve, ref_curve)
    assert ref_curve[-1] < ref_curve[0] and curve[-1] < curve[0]
    got_p = dict(m.named_parameters())
    worst = max(rel(got_p[k], ref_p[k]) for k in names)
    assert worst < 2e-3, worst


def test_save_model_round_trip(tmp_path):
    dims, params, *_ = _tiny_case()
    m = _engine(dims, params)
    path = tmp_path / "ckpt.pt"
    model_utils.save_model(m, str(path))
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"model_state_dict", "dims"} and ck["dims"]["n_audio_state"] == 384
    assert all(v.dtype == torch.float16 for k, v in ck["model_state_dict"].items() if v.is_floating_point())
    assert set(ck["model_state_dict"]) == set(params)


def test_fused_adamw_update_reaches_the_bf16_weight_shadows():
    """Regression: torch.optim.AdamW(fused=True) leaves tensor._version untouched; the forward after a step
    must nevertheless run on the UPDATED weights (bf16 shadows rebuilt), as the reference's autocast does."""
    dims, params, audio, y_in, y_out = _tiny_case()
    m = _engine(dims, params).train()
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
    w = m.encoder.blocks[0].mlp[0].weight
    m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.0).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    m.eval()
    with torch.no_grad():
        after = m(mel, y_in.to(DEV))
        fresh = _engine(dims, {k: v.detach().cpu() for k, v in m.state_dict().items()}).eval()
        want = fresh(mel, y_in.to(DEV))
    assert (w.detach().cpu() - params["encoder.blocks.0.mlp.0.weight"]).abs().max() > 1e-3  # the step moved the weights
    assert torch.equal(after, want)  # same kernels, same fp32 masters -> identical logits


def _grad_check(m, p_req, tag=""):
    errs = {n: rel(p.grad, p_req[n].grad) for n, p in m.named_parameters() if p.grad is not None and p_req[n].grad is not None}
    assert_grad_errors(errs, tag)
    return errs


def test_base_full_finetune_step_matches_oracle():
    """BASELINE configs[1] shape (whisper-base, full fine-tune, bf16; 2 of the 8 clips to keep the CPU oracle in seconds):
    GPU log-mel + SpecAugment-free forward, fused label-smoothed CE, backward vs the fp32 oracle."""
    dims = O.DIMS["base"]
    params = O.init_params(dims, seed=3)
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 32)
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    loss_ref = O.cross_entropy(O.Oracle(dims, p_req).forward(mel_ref, y_in), y_out, 0.1)
    loss_ref.backward()
    m = _engine(dims, params).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * loss_ref.item()
    _grad_check(m, p_req, "base vs fp32 oracle")


def test_turbo_lora_prompt_and_timestamp_targets_match_oracle():
    """BASELINE configs[4] shape: large-v3-turbo (32 encoder / 4 decoder layers, 128 mels), LoRA r=16 alpha=32 on every
    Linear, a batch item with a prompt (targets -100 up to and including the prompt) and timestamp tokens in the target
    stream.  One clip, S = 48, against the fp32 oracle with the same adapters (minLoRA parametrization form)."""
    dims = O.DIMS["large-v3-turbo"]
    params = O.init_params(dims, seed=5)
    torch.manual_seed(5)  # lora_A's kaiming init (global generator): the same adapters in every run
    g = torch.Generator().manual_seed(11)
    audio = torch.randn(1, 480000, generator=g) * 0.1
    sot_prev, sot, lang, task, ts0 = 50362, 50258, 50261, 50360, 50365
    prompt = torch.randint(0, 50257, (20,), generator=g)
    body = torch.randint(0, 50257, (22,), generator=g)
    body[::6] = ts0 + torch.randint(0, 1500, (len(body[::6]),), generator=g)  # timestamp tokens inside the text
    y_in = torch.cat([torch.tensor([sot_prev]), prompt, torch.tensor([sot, lang, task]), body]).unsqueeze(0)
    y_out = torch.cat([y_in[0, 1:], torch.tensor([50257])]).unsqueeze(0).clone()
    y_out[0, : 1 + len(prompt)] = -100  # no loss on the prompt (data_loader.py:322-340)
    assert y_in.shape[1] == 46
    r, alpha = 16, 32
    m = _engine(dims, params).train()
    lora_mod.apply_lora(m, {"rank": r, "lora_alpha": alpha, "lora_dropout": 0.0})
    ad = {}
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            if "lora_" in n:
                ad[n] = p.detach().cpu().clone()
    # oracle with the same adapters folded as W + (alpha/r) B A, gradients w.r.t. A and B through that form
    req = {n: v.clone().requires_grad_(True) for n, v in ad.items()}
    eff = dict(params)
    for n in [k for k in req if k.endswith("lora_A")]:
        base = n.replace(".parametrizations.weight.0.lora_A", ".weight")
        eff[base] = params[base] + (alpha / r) * req[n.replace("lora_A", "lora_B")] @ req[n]
    mel_ref = O.log_mel_spectrogram(audio, dims.n_mels)
    loss_ref = O.cross_entropy(O.Oracle(dims, eff).forward(mel_ref, y_in), y_out, 0.1)
    loss_ref.backward()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * loss_ref.item()
    named = dict(m.named_parameters())
    assert all(p.grad is None for n, p in named.items() if "lora" not in n)  # base weights frozen
    assert_lora_grad_errors({n: rel(named[n].grad, req[n].grad) for n in req})


def test_lora_dropout_masks_are_redrawn_per_micro_batch():
    """Two forward/backward passes with lora_dropout = 0.5 and NO optimizer step in between (an accumulation window):
    the second pass must run on ITS OWN mask.  The freshly drawn mask usually lands on the address of the freed first
    one, so the shadow caches must not key on (data_ptr, _version) — every draw carries a serial number
    (engine/ops.LoraSpec.draw_id).  dA / dB of the accumulated window against the oracle's W + s*B@(A*mask)."""
    from whisper_finetune.engine.whisper_model import Linear
    g = torch.Generator().manual_seed(0)
    lin = Linear(256, 384)
    lora_mod.add_lora_to_linear(lin, 8, 16, 0.5)
    lin.to(DEV).train()
    ad = lin.parametrizations.weight[0]
    with torch.no_grad():
        ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=g).to(DEV) * 0.1)
    lin.parametrizations.weight.original.requires_grad_(False)
    seen = []
    real_draw = ad.draw_mask

    def spy(training):
        mk = real_draw(training)
        seen.append(mk.detach().cpu().clone())
        return mk

    ad.draw_mask = spy
    xs = [torch.randn(64, 256, generator=g) for _ in range(2)]
    ws = [torch.randn(64, 384, generator=g) for _ in range(2)]
    ptrs = []
    for x, w in zip(xs, ws):
        y = lin(x.to(DEV))
        (y.float() * w.to(DEV)).sum().backward()
        ptrs.append(seen[-1].data_ptr())
        del y
    assert len(seen) == 2 and not torch.equal(seen[0], seen[1])
    # oracle: parametrization form with the two masks, gradients accumulated over the window
    A = ad.lora_A.detach().cpu().clone().requires_grad_(True)
    Bm = ad.lora_B.detach().cpu().clone().requires_grad_(True)
    W0 = lin.parametrizations.weight.original.detach().cpu()
    for x, w, mk in zip(xs, ws, seen):
        W = O.lora_effective_weight(W0, A, Bm, ad.scaling, mk)
        y = x.to(torch.bfloat16).float() @ W.T + lin.bias.detach().cpu()
        (y * w).sum().backward()
    assert rel(ad.lora_A.grad, A.grad) < 3e-2, rel(ad.lora_A.grad, A.grad)
    assert rel(ad.lora_B.grad, Bm.grad) < 3e-2, rel(ad.lora_B.grad, Bm.grad)
    # a stale first mask in the second pass would zero different columns of dA: check the support too
    dead = (seen[0] == 0) & (seen[1] == 0)
    assert ad.lora_A.grad[:, dead[0].to(DEV)].abs().max() == 0


def test_lora_masks_survive_a_second_forward_before_the_backward():
    """the mask pool's buffer is overwritten in place by the next draw.  `loss = model(a) + model(b)` runs two
    training forwards before the first backward: the first forward's adapters must back-propagate through THEIR masks (the pool
    moves still-alive specs to a snapshot).  Reference: the same two forwards run one after the other (forward, backward,
    forward, backward) under the same RNG seeds — the gradients of the accumulation window must agree.  Also: a train-mode call
    that enters below the root (`model.encoder(x)`) draws its own masks."""
    dims, params, audio, y_in, y_out = _tiny_case()
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    y_in, y_out = y_in.to(DEV), y_out.to(DEV)
    mel_b = torch.roll(mel, 1, 0).contiguous()

    def build():
        torch.manual_seed(5)
        m = Whisper(MODEL_DIMS["tiny"]); m.load_state_dict(params)
        lora_mod.apply_lora(m, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.5})
        gl = torch.Generator().manual_seed(9)
        for mod in m.modules():
            if "parametrizations" in mod._modules:
                ad = mod.parametrizations.weight[0]
                with torch.no_grad():
                    ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=gl) * 0.05)
        return m.to(DEV).train()

    def grads(m):
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad}

    m = build()
    torch.manual_seed(101); la = m(mel, y_in, targets=y_out)
    torch.manual_seed(202); lb = m(mel_b, y_in, targets=y_out)  # second draw while the first forward's graph is alive
    (la + lb).backward()
    g_joint = grads(m)
    m2 = build()
    torch.manual_seed(101); la2 = m2(mel, y_in, targets=y_out); la2.backward()
    torch.manual_seed(202); lb2 = m2(mel_b, y_in, targets=y_out); lb2.backward()
    g_seq = grads(m2)
    assert torch.equal(la, la2) and torch.equal(lb, lb2)
    worst = max(rel(g_joint[n], g_seq[n]) for n in g_seq)
    assert worst < 1e-5, worst  # (same kernels, same masks; only the order in which the two graphs add into .grad differs)
    # entering below the root: every call is its own draw
    pool = m.__dict__["_wft_lora_pool"]
    s0 = pool.serial
    with torch.no_grad():
        m.encoder(mel)
        s1 = pool.serial
        m.encoder(mel)
    assert s1 > s0 and pool.serial > s1
    # forward_loss called directly is ONE forward of the model — one draw, and the same loss / gradients as
    # the same call through the module
    m3, m4 = build(), build()
    pool3 = m3.__dict__["_wft_lora_pool"]
    draws = []
    on_forward = pool3._on_forward
    pool3._on_forward = lambda *a: (draws.append(1), on_forward(*a))[1]
    torch.manual_seed(303); l3 = m3.forward_loss(mel, y_in, y_out)
    l3.backward()
    assert len(draws) == 1
    torch.manual_seed(303); l4 = m4(mel, y_in, targets=y_out); l4.backward()
    assert torch.equal(l3, l4)
    g3, g4 = grads(m3), grads(m4)
    assert max(rel(g3[n], g4[n]) for n in g4) < 1e-6


def test_train_step_on_the_engine_follows_the_references_loss_sequence():
    """GPU leg of tests/golden/ref_train_step.npz (the REFERENCE'S OWN train_step driving the fp32 oracle, 4 optimizer steps x
    2 micro-batches, AdamW, clip 0.5, linear schedule): the engine in bf16 under this package's train_step stays within 3e-3
    of every loss of the sequence."""
    from pathlib import Path
    from tests.golden.gen_golden import TRAIN_STEP_CFG, TRAIN_STEP_OPT, train_step_case
    from whisper_finetune.model.optimizer import WftAdamW
    from whisper_finetune.model.scheduler import get_scheduler

    ref = np.load(Path(__file__).parent / "golden" / "ref_train_step.npz")
    m = _engine(ARCH_DIMS, arch_params(ARCH_DIMS, seed=3))
    opt = WftAdamW(m.parameters(), **TRAIN_STEP_OPT)
    sched = get_scheduler(opt, {"type": "linear", "warmup_steps": 2}, 4)
    it = iter(train_step_case())
    cfg = {**TRAIN_STEP_CFG, "mixed_precision_training": True}
    losses = [model_utils.train_step(m, it, opt, sched, dict(cfg), step=s) for s in range(1, 5)]
    np.testing.assert_allclose(losses, ref["losses"], rtol=3e-3)
    for n, p in m.named_parameters():
        want = float(ref["final_norm/" + n])
        assert abs(p.detach().float().norm().item() - want) < 2e-3 * want + 1e-6, n


def test_engine_refuses_fp32_and_fp16_requests():
    """training.mixed_precision_training: False / mp_dtype: fp16 must not silently compute bf16 (reference:
    model/model_utils.py:37-48,64 runs true fp32 / fp16 autocast there)."""
    dims, params, audio, y_in, y_out = _tiny_case(B=1, S=8)
    m = _engine(dims, params)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    base = {"accum_grad_steps": 1, "max_grad_norm": 1.0, "label_smoothing": 0.0}
    with pytest.raises(ValueError, match="fp32 compute"):
        model_utils.train_step(m, iter([(mel, y_in, y_out)]), opt, sched, {**base, "mixed_precision_training": False, "mp_dtype": "bf16"})
    with pytest.raises(ValueError, match="fp16"):
        model_utils.train_step(m, iter([(mel, y_in, y_out)]), opt, sched, {**base, "mixed_precision_training": True, "mp_dtype": "fp16"},
                               scaler=torch.amp.GradScaler("cuda"))


def test_per_clip_mel_path_equals_the_batched_front_end():
    """AudioDataset._calculate_mel (the reference's per-clip form, data/data_loader.py:273-292) and GpuMelLoader's batched form
    run the same kernels: identical draws -> identical mels (bitwise), including the partial-segment cut + minimum pad."""
    from whisper_finetune.data.data_loader import AudioDataset, GpuMelLoader, SimpleTokenizer, collate_fn
    from whisper_finetune.data.gpu_frontend import GpuFrontend

    class Records:
        column_names = ["audio", "text", "language", "prompt"]

        def __init__(self):
            g = torch.Generator().manual_seed(8)
            texts = ["<|0.00|>abc<|3.00|><|3.00|>", "<|0.00|>ohne schnitt<|4.00|>", "kein zeitstempel"]
            self.rows = [{"audio": {"array": (torch.randn(16000 * (5 + 3 * i), generator=g) * 0.1).numpy()}, "text": t,
                          "language": "de", "prompt": ""} for i, t in enumerate(texts)]

        def __len__(self):
            return len(self.rows)

        def __getitem__(self, i):
            return self.rows[i]

    sa = {"time_mask_param": 100, "freq_mask_param": 27, "time_warp_w": 80, "p": 1.0}
    ex = {"low_freq_range": 10, "high_freq_range": 6}
    hu = Records()
    ds = AudioDataset(hu, SimpleTokenizer(), n_mels=80, device=DEV, no_timestamp_training=True, spec_augment=True, spec_augment_params=sa,
                      extremes_spec_augment=True, extremes_spec_augment_params=ex, prompt_use_rate=0.0)
    torch.manual_seed(3)
    items = [ds[i] for i in range(3)]
    assert [int(it[5]) for it in items] == [300, 3000, 3000]  # cut at 3.00 s = 300 frames for the partial segment
    fe = GpuFrontend(80, DEV, True, sa, True, ex)
    batched = next(iter(GpuMelLoader([collate_fn(items)], fe, training_aug=True)))[0]
    torch.manual_seed(3)
    for i in range(3):
        rec = hu[i]
        torch.rand(1)  # the prompt-use draw of __getitem__ (no_timestamp_training: no no-timestamps draw)
        audio = np.pad(rec["audio"]["array"], (0, 480000 - rec["audio"]["array"].shape[0]))
        _, seg = ds._get_text_tokens(rec["text"], True)
        mel = ds._calculate_mel(audio, seg, True)
        assert mel.shape == (80, 3000)
        assert torch.equal(mel, batched[i]), i


def test_two_engine_models_interleaved():
    """A student / teacher (or EMA) setup: two engine models and two optimizers in one process, forwards interleaved.  The
    bf16 weight shadows are per module, the shadow epoch is bumped by ANY optimizer step (conservative: at worst a rebuild),
    and the fused bias-gradient column sums are keyed by tensors that stay alive until consumed: the student's step must equal
    the same step run alone, bit for bit."""
    dims, params, audio, y_in, y_out = _tiny_case(B=2, S=12)
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    yi, yo = y_in.to(DEV), y_out.to(DEV)

    def run(with_teacher):
        student = _engine(dims, params).train()
        opt = torch.optim.AdamW(student.parameters(), lr=1e-3, fused=True)
        teacher = t_opt = None
        if with_teacher:
            teacher = _engine(dims, {k: v * 1.01 for k, v in params.items()}).train()
            t_opt = torch.optim.SGD(teacher.parameters(), lr=1e-3)
        out = []
        for _ in range(2):
            if with_teacher:
                with torch.no_grad():
                    teacher(mel, yi)  # a forward of the other model before the student's
            loss = student(mel, yi, targets=yo, label_smoothing=0.1)
            if with_teacher:
                t_loss = teacher(mel, yi, targets=yo, label_smoothing=0.0)  # ... and one between its forward and backward
                t_loss.backward()
                t_opt.step(); t_opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step(); opt.zero_grad(set_to_none=True)
            out.append(loss.item())
        return out, {n: p.detach().clone() for n, p in student.named_parameters()}

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1
    assert all(torch.equal(p0[n], p1[n]) for n in p0)


@pytest.mark.parametrize("B,S", [(1, 448), (1, 1), (3, 5)])
def test_decoder_length_edge_cases_match_oracle(B, S):
    """Maximum decoder context (S = 448 = n_text_ctx, the reference truncates prompts to reach it: data_loader.py:331-338), a single
    token, and an odd ragged shape — bf16 engine and fp32 mode against the oracle."""
    dims, params, _, _, _ = _tiny_case()
    audio, y_in, y_out = O.synthetic_batch(dims, B, max(S, 5))
    y_in, y_out = y_in[:, :S].contiguous(), y_out[:, :S].contiguous()
    if S > 8:
        y_out[0, :4] = -100
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    logits_ref = O.Oracle(dims, p_req).forward(mel, y_in)
    loss_ref = O.cross_entropy(logits_ref, y_out, 0.1)
    loss_ref.backward()
    for mode, ltol, gtol in (("bf16", 1e-3, 0.12), ("fp32", 1e-4, 3e-3)):
        m = _engine(dims, params).set_compute_dtype(mode).train()
        loss = m(mel.to(DEV), y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
        loss.backward()
        assert abs(loss.item() - loss_ref.item()) < ltol * loss_ref.item(), (mode, loss.item(), loss_ref.item())
        errs = {n: rel(p.grad, p_req[n].grad) for n, p in m.named_parameters()}
        assert float(np.median(list(errs.values()))) < gtol / 4, mode
        if S > 1:  # with a single token the causal q / k gradients are exactly zero on both sides
            assert max(errs.values()) < gtol, (mode, sorted(errs.items(), key=lambda kv: -kv[1])[:3])
        m.eval()
        with torch.no_grad():
            logits = m(mel.to(DEV), y_in.to(DEV))
        assert logits.shape == (B, S, dims.n_vocab) and rel(logits, logits_ref) < (2e-2 if mode == "bf16" else 1e-3)


def test_lora_batched_refresh_is_bit_identical_to_the_per_linear_kernels():
    """One wft_lora_refresh_mt launch per training forward (merged shadows, their transposes and the rank-r gradient-GEMM
    operands of every adapter; engine/ops.LoraRefreshPlan, driven by the mask pool) against the per-Linear wft_lora_merge /
    wft_lora_pack path it replaces: three optimizer steps with dropout 0.25 and a two-micro-batch accumulation window give
    bit-identical losses, gradients and parameters; from the second forward on no per-Linear merge or pack is launched."""
    dims, params, audio, y_in, y_out = _tiny_case()
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    y_in, y_out = y_in.to(DEV), y_out.to(DEV)

    def run(batched: bool):
        torch.manual_seed(5)  # lora_A's kaiming init
        m = Whisper(MODEL_DIMS["tiny"]); m.load_state_dict(params)
        lora_mod.apply_lora(m, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.25})
        gl = torch.Generator().manual_seed(9)
        for mod in m.modules():
            if "parametrizations" in mod._modules:
                ad = mod.parametrizations.weight[0]
                with torch.no_grad():
                    ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=gl) * 0.05)
        m.to(DEV).train()
        pool = m.__dict__["_wft_lora_pool"]
        if not batched:
            pool.plan = None
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.05)
        counts = {"merge": 0, "pack": 0, "mt": 0}
        real = (K.lora_merge, K.lora_pack, K.lora_refresh_mt)

        def wrap(name, f):
            def g(*a, **k):
                counts[name] += 1
                return f(*a, **k)
            return g

        K.lora_merge, K.lora_pack, K.lora_refresh_mt = wrap("merge", real[0]), wrap("pack", real[1]), wrap("mt", real[2])
        try:
            torch.manual_seed(77)
            out, per_step = [], []
            for step in range(3):
                before = dict(counts)
                for micro in range(2):
                    loss = m(mel, y_in, targets=y_out, label_smoothing=0.1) / 2
                    loss.backward()
                    out.append(loss.detach().clone())
                grads = [p.grad.detach().clone() for p in m.parameters() if p.requires_grad]
                opt.step(); opt.zero_grad(set_to_none=True)
                per_step.append({k: counts[k] - before[k] for k in counts})
            return out, grads, [p.detach().clone() for p in m.parameters() if p.requires_grad], per_step
        finally:
            K.lora_merge, K.lora_pack, K.lora_refresh_mt = real

    l1, g1, p1, c1 = run(True)
    l0, g0, p0, c0 = run(False)
    assert all(torch.equal(a, b) for a, b in zip(l1, l0)), ([x.item() for x in l1], [x.item() for x in l0])
    assert all(torch.equal(a, b) for a, b in zip(g1, g0)) and all(torch.equal(a, b) for a, b in zip(p1, p0))
    n_ad = 4 * 6 + 4 * 10
    assert c0[1] == {"merge": 2 * n_ad, "pack": 2 * n_ad, "mt": 0}, c0
    assert c1[0]["merge"] >= n_ad and c1[1] == {"merge": 0, "pack": 0, "mt": 2} and c1[2] == {"merge": 0, "pack": 0, "mt": 2}, c1


def test_batched_weight_shadow_refresh_is_bit_identical_to_the_per_weight_kernel():
    """Full fine-tune: after an optimizer step ONE launch re-casts every trainable Linear group's bf16 shadow and its transpose
    (engine/ops.PlainRefreshPlan, the rank-0 rows of wft_lora_refresh_mt) instead of one wft_cast_pad_transpose per weight: three
    AdamW steps give bit-identical losses and parameters with and without it, and from the second step on the per-weight kernel
    only runs for the shapes the batched one does not take (the conv stem and the tied vocabulary matrix keep their own paths)."""
    from whisper_finetune.engine import ops
    from whisper_finetune.model.optimizer import WftAdamW
    dims, params, audio, y_in, y_out = _tiny_case()
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(DEV)
    y_in, y_out = y_in.to(DEV), y_out.to(DEV)

    def run(batched: bool):
        m = Whisper(MODEL_DIMS["tiny"]); m.load_state_dict(params)
        m.to(DEV).train()
        opt = WftAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        counts = {"cast": 0, "mt": 0, "bias": 0}
        real = (K.weight_shadow, K.lora_refresh_mt, K.mt_copy_f32)

        def wrap(name, f):
            def g(*a, **k):
                counts[name] += 1
                return f(*a, **k)
            return g

        K.weight_shadow, K.lora_refresh_mt, K.mt_copy_f32 = wrap("cast", real[0]), wrap("mt", real[1]), wrap("bias", real[2])
        old = ops._SHADOW_BATCH
        ops._SHADOW_BATCH = batched
        try:
            losses, per_step = [], []
            for step in range(3):
                before = dict(counts)
                loss = m(mel, y_in, targets=y_out, label_smoothing=0.1)
                loss.backward()
                opt.step(); opt.zero_grad(set_to_none=True)
                losses.append(loss.detach().clone())
                per_step.append({k: counts[k] - before[k] for k in counts})
            # the stacked bias vector of every fused group (q | k: none | v) is what its parameters say, after the last step too
            m.eval()
            with torch.no_grad():
                m(mel, y_in)
            for blk in list(m.encoder.blocks) + list(m.decoder.blocks):
                g, d = blk.attn._qkv_group, blk.attn.query.weight.shape[0]
                # (the q slice carries the softmax scale * log2(e): ops.QK_PRESCALE)
                assert torch.equal(g.bias[:d], blk.attn.query.bias.detach() * g.fwd_scale(0)) and torch.equal(g.bias[2 * d:3 * d], blk.attn.value.bias.detach())
                assert not g.bias[d:2 * d].any()
            for blk in m.decoder.blocks:
                g, d = blk.cross_attn._kv_group, blk.cross_attn.key.weight.shape[0]
                assert torch.equal(g.bias[d:2 * d], blk.cross_attn.value.bias.detach()) and not g.bias[:d].any()
            return losses, [p.detach().clone() for p in m.parameters()], per_step
        finally:
            K.weight_shadow, K.lora_refresh_mt, K.mt_copy_f32 = real
            ops._SHADOW_BATCH = old

    l1, p1, c1 = run(True)
    l0, p0, c0 = run(False)
    assert all(torch.equal(a, b) for a, b in zip(l1, l0)) and all(torch.equal(a, b) for a, b in zip(p1, p0))
    assert c0[1]["mt"] == 0 and c0[1]["cast"] >= 4 * 6 + 4 * 10
    assert c1[1]["mt"] == 1 and c1[2]["mt"] == 1 and c1[1]["cast"] <= c0[1]["cast"] - (4 * 6 + 4 * 10) + 4, (c1, c0)
    # ... and ONE wft_mt_copy_f32 launch restacks the q / v (and cross k / v) bias vectors of all fused groups
    assert c0[1]["bias"] == 0 and c1[1]["bias"] == 1 and c1[2]["bias"] == 1, (c1, c0)


def test_encoder_output_gradient_accumulated_in_the_gemm_epilogues():
    """ops.GradAccum: the key / value projections of all decoder blocks add their input gradients inside their backward-data GEMMs
    (whisper's MultiHeadAttention.forward with `xa`, reached from the reference's model_utils.py:320-322).  Same sum as autograd's
    pairwise adds up to one bf16 rounding per block: every encoder-side gradient agrees with the autograd-summed run, with stochastic
    depth (skipped decoder blocks never register) and under checkpoint recompute (a recomputed forward must not count twice)."""
    from whisper_finetune.engine import whisper_model as WM
    from whisper_finetune.model.model_utils import CheckpointedStochasticTextDecoder
    dims, params, audio, y_in, y_out = _tiny_case()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))

    def grads(accum: bool, sd: float, recompute: bool):
        old = WM._XA_ACCUM
        WM._XA_ACCUM = accum
        try:
            m = Whisper(MODEL_DIMS["tiny"])
            if sd > 0 or recompute:
                m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                              dims.n_text_layer, sd)
                m.decoder.recompute = recompute
            m.load_state_dict(params)
            m.to(DEV).train()
            torch.manual_seed(123)  # the same skip draws in both runs
            loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1)
            loss.backward()
            return loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            WM._XA_ACCUM = old

    for sd, recompute in ((0.0, False), (0.5, False), (0.0, True), (0.5, True)):
        l0, g0 = grads(False, sd, recompute)
        l1, g1 = grads(True, sd, recompute)
        assert l0 == l1
        assert g0.keys() == g1.keys()
        enc = [n for n in g0 if n.startswith("encoder.")]
        assert enc and all(g0[n].abs().sum() > 0 for n in enc if "blocks.0.attn.query.weight" in n)
        for n in g0:
            assert rel(g1[n], g0[n]) < (1e-2 if n.startswith("encoder.") else 1e-6), (n, sd, recompute, rel(g1[n], g0[n]))


def test_encoder_output_gradient_fork_survives_two_decoder_passes_and_partial_backwards():
    """ADVICE r4 on ops.GradAccum: (1) two decoder passes over ONE encoder output with different stochastic-depth skips (31 + 31
    arrivals against 32 registered ids used to hand the sum over early and drop the rest); (2) `autograd.grad(inputs=subset)` in
    front of the real backward used to leave a count behind.  With the fork node both give the encoder the gradient autograd's own
    summation gives (WFT_XA_ACCUM=0 twin, same skip draws)."""
    from whisper_finetune.engine import whisper_model as WM
    from whisper_finetune.model.model_utils import CheckpointedStochasticTextDecoder
    dims, params, audio, y_in, y_out = _tiny_case()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))

    def run(accum: bool):
        old = WM._XA_ACCUM
        WM._XA_ACCUM = accum
        try:
            m = Whisper(MODEL_DIMS["tiny"])
            m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                          dims.n_text_layer, 0.5)
            m.decoder.recompute = False
            m.load_state_dict(params)
            m.to(DEV).train()
            torch.manual_seed(7)
            xa = m.encoder(mel)
            la = m.decoder(y_in.to(DEV), xa).float()          # two passes of the decoder over one encoder output: the skip draws differ
            lb = m.decoder(y_in.flip(1).to(DEV), xa).float()
            loss = la.square().mean() + lb.square().mean()
            # a partial pass first: gradients of one decoder weight only — the fork node is not part of it
            w = m.decoder.blocks[0].cross_attn.key.weight
            (gw,) = torch.autograd.grad(loss, [w], retain_graph=True, allow_unused=True)
            loss.backward()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, gw
        finally:
            WM._XA_ACCUM = old

    g0, w0 = run(False)
    g1, w1 = run(True)
    assert g0.keys() == g1.keys() and any(n.startswith("encoder.") for n in g0)
    assert (w0 is None) == (w1 is None) and (w0 is None or rel(w1, w0) < 1e-6)
    for n in g0:
        assert rel(g1[n], g0[n]) < (1e-2 if n.startswith("encoder.") else 1e-6), (n, rel(g1[n], g0[n]))
