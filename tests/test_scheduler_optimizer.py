"""LR schedules against the reference's own get_scheduler output (tests/golden/ref_sched.json, generated in the build
container by tests/golden/gen_golden.py:gen_ref_sched from the reference's own model/scheduler.py), optimizer factory behaviour."""
import json
from pathlib import Path

import pytest
import torch

from whisper_finetune.model.optimizer import WftAdamW, get_optimizer
from whisper_finetune.model.scheduler import get_scheduler

GOLD = json.loads((Path(__file__).parent / "golden" / "ref_sched.json").read_text())


@pytest.mark.parametrize("kind", sorted(GOLD))
def test_schedule_matches_reference(kind):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    import random

    random.seed(0)  # the chill variant jitters with random.uniform: the table was taken under this seed
    s = get_scheduler(opt, GOLD[kind]["conf"], 120)
    for ref in GOLD[kind]["lrs"]:
        assert abs(opt.param_groups[0]["lr"] - ref) < 1e-12
        opt.step(); s.step()


def test_unknown_scheduler_raises():
    with pytest.raises(Exception, match="Unknown learning rate scheduler"):
        get_scheduler(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0), {"type": "nope", "warmup_steps": 1}, 10)


def test_get_optimizer_variants():
    m = torch.nn.Linear(4, 4)
    m.bias.requires_grad = False
    conf = {"type": "adamw", "8bit": False, "params": {"lr": 1e-3, "weight_decay": 0.1, "betas": [0.9, 0.98], "eps": 1e-6}}
    opt = get_optimizer(m, conf)
    assert isinstance(opt, torch.optim.AdamW) and len(opt.param_groups[0]["params"]) == 1  # only trainable parameters
    assert isinstance(get_optimizer(m, {**conf, "type": "adam"}), torch.optim.Adam)
    assert isinstance(get_optimizer(m, {**conf, "wft": True}), WftAdamW)
    with pytest.raises(ValueError):
        get_optimizer(m, {**conf, "type": "sgd"})
    with pytest.raises(ImportError):
        get_optimizer(m, {**conf, "8bit": True})


def test_muon_param_groups_match_the_reference():
    """get_optimizer's Muon branch (partition, RMS-matched lr / weight-decay groups, metadata, group keys) against the
    reference's own helper functions (tests/golden/ref_optim.json) and its tests/test_optimizer.py:22-60 assertions."""
    from tests.golden.gen_golden import MUON_CONF, fake_muon_model
    from whisper_finetune.model.optimizer import WftMuonWithAuxAdam, _use_muon_optimizer

    gold = json.loads((Path(__file__).parent / "golden" / "ref_optim.json").read_text())
    m = fake_muon_model()
    names = {id(p): n for n, p in m.named_parameters()}
    opt = get_optimizer(m, MUON_CONF)
    assert isinstance(opt, WftMuonWithAuxAdam)
    assert len(opt._lr_group_metadata) == len(opt.param_groups) >= 2
    muon_groups = [g for g in opt.param_groups if g["use_muon"]]
    aux_groups = [g for g in opt.param_groups if not g["use_muon"]]
    assert [names[id(p)] for g in muon_groups for p in g["params"]] == [n for g in gold["groups"]["True"] for n in g["names"]]
    assert sorted(n for g in muon_groups for n in (names[id(p)] for p in g["params"])) == sorted(gold["muon_names"])
    assert [names[id(p)] for p in aux_groups[0]["params"]] == gold["aux_names"]
    for g, ref in zip(muon_groups, gold["groups"]["True"]):
        assert set(g.keys()) == {"params", "lr", "momentum", "weight_decay", "use_muon"}
        assert g["lr"] == ref["lr"] and g["weight_decay"] == ref["weight_decay"] and g["momentum"] == ref["momentum"]
    for idx, g in enumerate(opt.param_groups):
        meta = opt._lr_group_metadata[idx]
        if g["use_muon"]:
            assert meta == {"lr_log_label": "muon", "base_lr_unscaled": MUON_CONF["muon_params"]["lr"]}
        else:
            assert set(g.keys()) == {"params", "lr", "betas", "eps", "weight_decay", "use_muon"}
            assert meta == {"lr_log_label": "aux_adamw", "base_lr_unscaled": MUON_CONF["params"]["lr"]}
    flat = get_optimizer(fake_muon_model(), {**MUON_CONF, "muon_match_adamw_update_rms": False})
    fg = [g for g in flat.param_groups if g["use_muon"]]
    assert len(fg) == 1 and fg[0]["lr"] == gold["groups"]["False"][0]["lr"] and fg[0]["weight_decay"] == gold["groups"]["False"][0]["weight_decay"]
    for conf, want in gold["use_muon"]:
        assert _use_muon_optimizer(conf) == want
    with pytest.raises(ValueError, match="muon_ndim_threshold"):
        get_optimizer(fake_muon_model(), {**MUON_CONF, "muon_ndim_threshold": 0})
    with pytest.raises(ValueError, match="muon_match_factor"):
        get_optimizer(fake_muon_model(), {**MUON_CONF, "muon_match_factor": 0})


def test_libwft_optimizers_refuse_cpu_parameters():
    from whisper_finetune.engine.lib import WftError

    p = torch.nn.Parameter(torch.randn(8, 8))
    p.grad = torch.randn(8, 8)
    with pytest.raises(WftError):
        WftAdamW([p]).step()


def test_oracle_newton_schulz_orthogonalises():
    """The restated zeropower_via_newtonschulz5 maps every singular value into Muon's [0.6, 1.25] band, wide or tall."""
    from oracle import whisper_oracle as O

    g = torch.Generator().manual_seed(0)
    for shape in ((48, 160), (160, 48), (64, 64)):
        x = O.zeropower_via_newtonschulz5(torch.randn(shape, generator=g)).float()
        sv = torch.linalg.svdvals(x)
        assert x.shape == shape and 0.55 < sv.min() and sv.max() < 1.3


@pytest.mark.gpu
def test_wft_adamw_fused_clip_matches_clip_grad_norm_then_torch_adamw():
    """train_step's clip_grad_norm_ + optimizer.step() (model_utils.py:107,122) == fuse_clip_grad_norm + one wft_mt_adamw."""
    torch.manual_seed(0)
    ws = [torch.randn(300, 77), torch.randn(1001), torch.randn(70001), torch.randn(3)]
    ref = [torch.nn.Parameter(w.clone().cuda()) for w in ws]
    mine = [torch.nn.Parameter(w.clone().cuda()) for w in ws]
    kw = dict(lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    a, b = torch.optim.AdamW(ref, **kw), WftAdamW(mine, **kw)
    for it in range(3):
        for r, m in zip(ref, mine):
            g = torch.randn_like(r) * (3.0 if it < 2 else 1e-4)  # clipped twice, then far below max_norm
            r.grad, m.grad = g.clone(), g.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        a.step()
        b.fuse_clip_grad_norm(1.0)
        b.step()
        assert abs(b.last_grad_norm.item() - want_norm.item()) < 1e-4 * want_norm.item()
        for r, m in zip(ref, mine):
            assert (r - m).abs().max().item() < 2e-6
    assert b.state[mine[0]]["step"] == 3 and set(b.state[mine[0]]) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.gpu
def test_wft_muon_with_aux_adam_matches_the_restated_package():
    """Tall, wide, square and padded (r = 16) matrices + an aux-Adam vector, three steps, against
    oracle.muon_with_aux_adam_step (bf16 Newton-Schulz on the CPU).  Tolerance: the orthogonalised update is a bf16
    quantity (8 mantissa bits) pushed through 15 chained bf16 GEMMs whose rounding order differs (one fused
    alpha*acc + beta*residual rounding here, three in torch).  Two bf16 evaluations of the iteration differ by 2-4 %
    relative L2 (measured: 2.0-3.8 %, the same distance the CPU bf16 restatement keeps from an fp32 evaluation, checked
    below), so the bound is 6e-2 on the Muon step; the aux-Adam step is fp32: 1e-5."""
    from oracle import whisper_oracle as O
    from whisper_finetune.engine import kernels as K
    from whisper_finetune.model.optimizer import WftMuonWithAuxAdam

    torch.manual_seed(1)
    shapes = [(128, 384), (128, 384), (384, 128), (256, 256), (16, 320), (320, 16)]
    ws = [torch.randn(s) * 0.05 for s in shapes]
    bias = torch.randn(77) * 0.1
    dev = torch.device("cuda:0")
    mine = [torch.nn.Parameter(w.clone().to(dev)) for w in ws] + [torch.nn.Parameter(bias.clone().to(dev))]
    groups = [{"params": mine[:-1], "use_muon": True, "lr": 3e-3, "momentum": 0.95, "weight_decay": 0.01},
              {"params": mine[-1:], "use_muon": False, "lr": 1e-3, "betas": (0.9, 0.98), "eps": 1e-6, "weight_decay": 0.01}]
    opt = WftMuonWithAuxAdam(groups)
    ref_p = [w.clone() for w in ws] + [bias.clone()]
    state = {}
    for it in range(3):
        grads = [torch.randn(p.shape) * 0.02 for p in ref_p]
        for p, g in zip(mine, grads):
            p.grad = g.clone().to(dev)
        before = [p.detach().clone() for p in mine]
        opt.step()
        ref_before = [p.clone() for p in ref_p]
        O.muon_with_aux_adam_step([{**groups[0], "params": list(zip(ref_p[:-1], [g.clone() for g in grads[:-1]]))},
                                   {**groups[1], "params": [(ref_p[-1], grads[-1].clone())]}], state)
        for i, (p, q) in enumerate(zip(mine, ref_p)):
            step_mine = (p.detach() - before[i]).cpu()
            step_ref = q - ref_before[i]
            rel = ((step_mine - step_ref).norm() / step_ref.norm()).item()
            assert rel < (6e-2 if i < len(shapes) else 1e-5), (it, i, tuple(q.shape), rel)
            assert (p.detach().cpu() - q).abs().max().item() < 2e-4
    # calibration of that bound: bf16 vs fp32 evaluation of the same iteration on the CPU
    gcal = torch.randn(128, 384) * 0.02
    x32 = gcal / (gcal.norm() + 1e-7)
    for _ in range(5):
        a32 = x32 @ x32.T
        x32 = O.NS_COEFFS[0] * x32 + (O.NS_COEFFS[1] * a32 + O.NS_COEFFS[2] * a32 @ a32) @ x32
    xb = O.zeropower_via_newtonschulz5(gcal).float()
    assert 5e-3 < ((xb - x32).norm() / x32.norm()).item() < 6e-2
    # the orthogonalised direction itself: singular values in Muon's band (wide matrix: a square Gaussian's smallest
    # singular values are ~0 and stay small through five iterations, in the reference as well)
    upd = K.muon_group_step([mine[0].data], [torch.randn(128, 384, device=dev)], [torch.zeros(128, 384, device=dev)], 0.0, 0.0, 0.95,
                            return_update=True)
    sv = torch.linalg.svdvals(upd[0].cpu())
    assert 0.55 < sv.min() and sv.max() < 1.3


@pytest.mark.gpu
def test_muon_parameter_without_a_gradient_steps_like_one_with_a_zero_gradient():
    """muon.py hands a parameter that took no gradient (a block stochastic depth skipped) torch.zeros_like(p); the optimizer here
    passes a NULL table row instead (no allocation, no fill): parameters and momentum buffers must come out bit-identical, with
    and without the fused clip, and the missing gradient must stay missing."""
    from whisper_finetune.model.optimizer import WftMuonWithAuxAdam

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(128, 384), (128, 384), (128, 384), (320, 16), (320, 16)]
    ws = [torch.randn(s) * 0.05 for s in shapes]
    runs = []
    for materialise in (False, True):
        ps = [torch.nn.Parameter(w.clone().to(dev)) for w in ws]
        opt = WftMuonWithAuxAdam([{"params": ps, "use_muon": True, "lr": 3e-3, "momentum": 0.95, "weight_decay": 0.01}])
        g = torch.Generator().manual_seed(5)
        for it in range(3):
            for i, p in enumerate(ps):
                grad = torch.randn(p.shape, generator=g) * 0.02
                missing = (i + it) % 3 == 1  # a different subset every step; momentum exists from earlier steps for some
                p.grad = (torch.zeros_like(p) if materialise else None) if missing else grad.to(dev)
            if it == 1:
                opt.fuse_clip_grad_norm(0.05)
            opt.step()
            if not materialise:
                assert [p.grad is None for p in ps] == [(i + it) % 3 == 1 for i in range(len(ps))]
        runs.append(([p.detach().clone() for p in ps], [opt.state[p]["momentum_buffer"].clone() for p in ps]))
    for a, b in zip(runs[0][0] + runs[0][1], runs[1][0] + runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_pointer_tables_reach_the_device_unchanged_and_staging_buffers_are_recycled():
    """kernels.upload_table: pinned staging + stream-ordered copy instead of torch.tensor(list, device=...) (which drains the
    stream and stalls the host, 13 times per Muon step before)."""
    from whisper_finetune.engine import kernels as K

    dev = torch.device("cuda:0")
    big = torch.empty(1 << 26, device=dev)
    outs = []
    for rep in range(40):
        big.normal_()  # keep the stream busy: the copies queue behind real work
        vals = [(rep * 1000003 + i * 7919) % (1 << 47) for i in range(1 + (rep * 37) % 700)]
        outs.append((vals, K.upload_table(vals, torch.int64, dev)))
        vals32 = [i - rep for i in range(300)]
        outs.append((vals32, K.upload_table(vals32, torch.int32, dev)))
    torch.cuda.synchronize()
    for vals, t in outs:
        assert t.tolist() == vals
    assert sum(len(v) for v in K._STAGER.pool.values()) <= 80  # (bounded: one staging buffer per copy still in flight)
    K.upload_table([1, 2, 3], torch.int64, dev)
    torch.cuda.synchronize()
    n_before = sum(len(v) for v in K._STAGER.pool.values())
    for _ in range(10):
        K.upload_table([4, 5, 6], torch.int64, dev)
        torch.cuda.synchronize()
    assert sum(len(v) for v in K._STAGER.pool.values()) == n_before  # an idle buffer is reused, not replaced


@pytest.mark.gpu
def test_single_tensor_adamw_entry_point_matches_torch_adamw_with_clip_scale():
    """wft_adamw_step (flat range, device-scalar gradient scale, bf16 shadow refreshed in the same pass)."""
    from whisper_finetune.engine import kernels as K

    torch.manual_seed(0)
    w = torch.randn(300, 77)
    ref = torch.nn.Parameter(w.clone().cuda())
    p = w.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.empty(p.shape, dtype=torch.bfloat16, device="cuda")
    a = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    gscale = torch.tensor([0.5], device="cuda")
    for step in range(1, 4):
        g = torch.randn_like(p)
        ref.grad = g * 0.5
        a.step()
        K.adamw_step(p, g, m, v, shadow, 1e-2, 0.9, 0.98, 1e-6, 0.1, 1 - 0.9 ** step, 1 - 0.98 ** step, gscale)
    torch.testing.assert_close(p, ref.detach(), atol=1e-6, rtol=1e-5)
    assert torch.equal(shadow, p.to(torch.bfloat16))


def test_muon_spectral_check_accepts_the_oracle_and_rejects_wrong_updates():
    """tests/_muon_spectral.py (the large-v3 Muon check) on the CPU: the oracle's bf16 Newton-Schulz update
    of rank-16 gradients with singular-value spreads of 1 ... 1e-3 passes against a second bf16 evaluation — also where two
    evaluations differ by 20 % in relative L2 — and so does the fp32 evaluation, while the negated, the un-scaled (rows > cols) and the un-normalised update each fail."""
    from oracle import whisper_oracle as O
    from tests._muon_spectral import NS_BAND, negative_controls, ns5_scalar, spectral_violations

    x = torch.logspace(-3, 0, 4000)
    y = ns5_scalar(x)[x >= 0.025]  # a strong direction (sigma >= 0.1 sigma_max) of a rank-16 matrix has sigma / |M|_F >= 0.025
    assert NS_BAND[0] < float(y.min()) and float(y.max()) < NS_BAND[1]
    g = torch.Generator().manual_seed(0)
    for shape in ((16, 1280), (1280, 16), (64, 64), (5120, 16)):
        r = min(shape)
        for spread in (1.0, 1e-1, 1e-2, 1e-3):
            P = torch.linalg.qr(torch.randn(shape[0], r, generator=g))[0]
            Q = torch.linalg.qr(torch.randn(shape[1], r, generator=g))[0]
            sv = torch.logspace(0, float(torch.log10(torch.tensor(spread))), r) * 3e-3
            M = (P * sv) @ Q.T
            sc = max(1, shape[0] / shape[1]) ** 0.5
            U16 = O.zeropower_via_newtonschulz5(M.clone()).float() * sc
            U32 = O.zeropower_via_newtonschulz5(M.clone(), dtype=torch.float32) * sc
            # a second bf16 evaluation of the same update (the iteration is scale invariant; other roundings): the reference the
            # GPU update is held against in tests/test_large_v3_gpu.py is such an evaluation, rounding noise included
            U16b = O.zeropower_via_newtonschulz5(M.clone() * 1.37).float() * sc
            assert spectral_violations(U16, M, U16b) == [], (shape, spread, spectral_violations(U16, M, U16b))
            assert spectral_violations(U32, M, U16b) == [], (shape, spread)
            ctl = negative_controls(U16, M)
            assert ("unscaled" in ctl) == (shape[0] > shape[1])
            for name, wrong in ctl.items():
                assert spectral_violations(wrong, M, U16b), (shape, spread, name)
