"""Host-side mirror of the reference interface (whisper_finetune.*): step arithmetic, train_step
control flow (DDP faked exactly as the reference's own tests do, tests/test_training_utils.py:400-533),
stochastic depth, deep-SpecAugment draws, SpecAugment draw order, LoRA module structure.  CPU only."""
import contextlib

import numpy as np
import pytest
import torch

import whisper_finetune.runtime as rt
from whisper_finetune import utils
from whisper_finetune.data.gpu_frontend import GpuFrontend, mel_filters
from whisper_finetune.engine.whisper_model import MODEL_DIMS, Linear, Whisper
from whisper_finetune.model import lora, model_utils


def test_step_tables_match_reference(golden_host):
    for n, world, bs, ep, acc, dl, ref in golden_host["train_steps_table"]:
        cfg = {"training": {"epochs": ep if ep != int(ep) else int(ep), "accum_grad_steps": int(acc)}, "dataset": {"batch_size": int(bs)}}
        assert utils.calculate_training_steps(cfg, range(int(n)), int(world), bool(dl)) == int(ref)
    for ts, ep, ev, ref in golden_host["val_steps_table"]:
        assert utils.calculate_val_steps({"training": {"train_steps": int(ts), "epochs": int(ep), "eval_steps": ev}}) == int(ref)
    for a, w, ref in golden_host["local_accum"]:
        assert utils.resolve_local_accum_grad_steps(int(a), int(w)) == int(ref)
    with pytest.raises(ValueError):
        utils.resolve_local_accum_grad_steps(8, 3)
    with pytest.raises(ValueError):
        utils.resolve_local_accum_grad_steps(0, 1)


def test_set_seed_reproducible():
    utils.set_seed(5); a = torch.rand(3)
    utils.set_seed(5); b = torch.rand(3)
    assert torch.equal(a, b)


class _TinyDDPModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 6)
        self.no_sync_entries = 0

    def forward(self, x, y_in):
        return self.lin(x).unsqueeze(1).expand(-1, y_in.shape[1], -1)

    @contextlib.contextmanager
    def no_sync(self):
        self.no_sync_entries += 1
        yield


def _batches(n):
    for _ in range(n):
        yield torch.randn(2, 4), torch.zeros(2, 3, dtype=torch.long), torch.randint(0, 6, (2, 3))


class _Sched:
    def __init__(self): self.steps = 0
    def step(self): self.steps += 1


T_CFG = {"mixed_precision_training": False, "accum_grad_steps": 4, "max_grad_norm": 1.0, "mp_dtype": "bf16"}


def test_train_step_enters_no_sync_on_all_but_last_microbatch(monkeypatch):
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", True)
    m, s = _TinyDDPModel(), _Sched()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    before = m.lin.weight.detach().clone()
    loss = model_utils.train_step(m, _batches(4), opt, s, dict(T_CFG))
    assert m.no_sync_entries == 3 and s.steps == 1
    assert isinstance(loss, float) and loss > 0
    assert not torch.equal(before, m.lin.weight)
    assert all(p.grad is None for p in m.parameters())  # zero_grad(set_to_none=True)


def test_train_step_switches_to_per_tile_launches_only_beside_the_gradient_exchange(monkeypatch):
    """runtime.exchange_launch_mode: in a multi-process job the LAST micro-batch of an accumulation window (the one whose gradients are
    all-reduced while its backward pass runs) is traced with `backward_launch_mode() == 1`, which the engine's autograd nodes note in
    their forward and hand to their backward kernels per call; everything else keeps the persistent grids.  The mode is THREAD-LOCAL
    (an evaluator on another thread is not affected) and nests / unwinds on errors; libwft itself holds no launch state."""
    import threading

    seen = []

    class Spy(_TinyDDPModel):
        def forward(self, x, y_in):
            seen.append(rt.backward_launch_mode())
            return super().forward(x, y_in)

    assert rt.backward_launch_mode() == 0
    with rt.exchange_launch_mode(True):
        assert rt.backward_launch_mode() == 1
        other = []
        t = threading.Thread(target=lambda: other.append(rt.backward_launch_mode()))
        t.start(); t.join()
        assert other == [0]  # another thread (an evaluator, a second model) keeps its own mode
        with rt.exchange_launch_mode(False):
            assert rt.backward_launch_mode() == 1
    with rt.exchange_launch_mode(False):
        assert rt.backward_launch_mode() == 0
    try:
        with rt.exchange_launch_mode(True):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert rt.backward_launch_mode() == 0
    # train_step: the mode is decided per micro-batch; on the CPU (this test) it never switches — the device check — so patch it
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", True)
    real = rt.exchange_launch_mode
    monkeypatch.setattr(rt, "exchange_launch_mode", lambda active: real(rt.IS_DISTRIBUTED and len(seen) == 3))
    m = Spy()
    model_utils.train_step(m, _batches(4), torch.optim.SGD(m.parameters(), lr=0.1), _Sched(), dict(T_CFG))
    assert seen == [0, 0, 0, 1]
    assert rt.backward_launch_mode() == 0


def test_train_step_no_sync_not_used_without_ddp(monkeypatch):
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", False)
    m = _TinyDDPModel()
    model_utils.train_step(m, _batches(4), torch.optim.SGD(m.parameters(), lr=0.1), _Sched(), dict(T_CFG))
    assert m.no_sync_entries == 0


def test_train_step_loss_is_sum_of_scaled_microbatch_means(monkeypatch):
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", False)
    torch.manual_seed(0)
    m = _TinyDDPModel()
    data = list(_batches(4))
    expect = sum(torch.nn.functional.cross_entropy(m(x, yi).transpose(1, 2), yo, label_smoothing=0.1).item() / 4 for x, yi, yo in data)
    got = model_utils.train_step(m, iter(data), torch.optim.SGD(m.parameters(), lr=0.0), _Sched(), {**T_CFG, "label_smoothing": 0.1})
    assert abs(got - expect) < 1e-6


def test_train_step_illegal_memory_reraised_under_ddp(monkeypatch):
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", True)
    calls = {"n": 0}

    class Bad(_TinyDDPModel):
        def forward(self, x, y_in):
            calls["n"] += 1
            raise RuntimeError("CUDA error: an illegal memory access was encountered")

    m = Bad()
    with pytest.raises(RuntimeError, match="illegal memory"):
        model_utils.train_step(m, _batches(8), torch.optim.SGD(m.parameters(), lr=0.1), _Sched(), dict(T_CFG))
    assert calls["n"] == 1  # no retry under DDP
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", False)
    calls["n"] = 0
    with pytest.raises(RuntimeError):
        model_utils.train_step(m, _batches(8), torch.optim.SGD(m.parameters(), lr=0.1), _Sched(), dict(T_CFG))
    assert calls["n"] == 3  # three attempts on a single GPU


def test_train_step_fp16_needs_scaler():
    m = _TinyDDPModel()
    with pytest.raises(ValueError, match="GradScaler"):
        model_utils.train_step(m, _batches(1), torch.optim.SGD(m.parameters(), lr=0.1), _Sched(),
                               {**T_CFG, "mixed_precision_training": True, "mp_dtype": "fp16"})


def test_infinite_iter_cycles_and_sets_epoch():
    class S:
        def __init__(self): self.epochs = []
        def set_epoch(self, e): self.epochs.append(e)

    class DL(list):
        pass

    dl = DL([1, 2]); dl.sampler = S()
    it = model_utils.infinite_iter(dl)
    assert [next(it) for _ in range(5)] == [1, 2, 1, 2, 1]
    assert dl.sampler.epochs == [0, 1, 2]


def test_stochastic_depth_matches_reference(golden_host):
    class SD(model_utils.StochasticDepthMixin, torch.nn.Module):
        pass
    sd = SD().train()
    x = torch.tensor([[1.0, -2.0, 3.0]])
    torch.manual_seed(0)
    for ref in golden_host["sd_outs"]:
        np.testing.assert_allclose(sd.stochastic_depth(x, lambda t: t * 2 + 1, 0.4).detach().numpy(), ref, rtol=1e-6)
    sd.eval()
    np.testing.assert_allclose(sd.stochastic_depth(x, lambda t: t * 2 + 1, 0.4).detach().numpy(), golden_host["sd_eval"], rtol=1e-6)
    sd.train()
    assert torch.equal(sd.stochastic_depth(x, lambda t: t * 2 + 1, 1.0), x)  # keep_prob 0 or skipped: identity


def test_resample_block_list():
    blocks = torch.nn.ModuleList([torch.nn.Linear(1, 1) for _ in range(4)])
    assert len(model_utils._resample_block_list(blocks, 2)) == 2
    up = model_utils._resample_block_list(blocks, 6)
    assert len(up) == 6 and up[0] is blocks[0]
    with pytest.raises(ValueError):
        model_utils._resample_block_list(blocks, 0)


def _tiny():
    return Whisper(MODEL_DIMS["tiny"])


def test_deep_spec_augment_draws_match_reference(golden_host):
    """Same torch.manual_seed -> same zeroed time rows / channels as the reference's hooks produced."""
    from whisper_finetune.engine.whisper_model import ModelDimensions
    m = Whisper(ModelDimensions(80, 150, 128, 2, 3, 100, 16, 128, 2, 1)).train()
    model_utils.register_deep_spec_augment_hooks(m, time_mask_param=30, freq_mask_param=20, p=1.0)
    torch.manual_seed(42)
    for hook in m.encoder._forward_pre_hooks.values():
        hook(m.encoder, None)
    for i, blk in enumerate(m.encoder.blocks):
        drawer = blk.attn_ln.deep_spec_augment
        if i == 2:
            assert drawer is None  # last block never augmented
            continue
        t0, t1, c0, c1 = drawer()
        rows = np.zeros(150, bool); rows[t0:t1] = True
        cols = np.zeros(128, bool); cols[c0:c1] = True
        np.testing.assert_array_equal(rows, golden_host[f"dsa_rows{i}"])
        np.testing.assert_array_equal(cols, golden_host[f"dsa_cols{i}"])
    with pytest.raises(ValueError):
        model_utils.register_deep_spec_augment_hooks(m, 1, 1, p=1.5)


def test_gpu_frontend_draw_order_is_the_references():
    """warp randint x2 -> time rand x2 -> freq rand x2 per clip (data_loader.py:284-287, data/utils.py:107,111)."""
    fe = GpuFrontend.__new__(GpuFrontend)
    fe.n_mels, fe.spec_augment, fe.p, fe.extremes = 80, True, 1.0, True
    fe.time_mask_param, fe.freq_mask_param, fe.time_warp_w = 100, 27, 80
    fe.low_freq_range, fe.high_freq_range = 10, 6
    torch.manual_seed(123)
    params, ext = fe.draw(2)
    torch.manual_seed(123)
    for b in range(2):
        wp = int(torch.randint(80, 3000 - 80, (1,))); wd = int(torch.randint(-80, 80, (1,)))
        v = torch.rand(1) * 100; mn = torch.rand(1) * (3000 - v); t0 = int(mn.long()); t1 = t0 + int(v.long())
        v = torch.rand(1) * 27; mn = torch.rand(1) * (80 - v); f0 = int(mn.long()); f1 = f0 + int(v.long())
        r = torch.rand(1).item()
        assert params[b].tolist() == [1, wp, wd, t0, t1, f0, f1, 0]
        assert ext[b].tolist() == [int(round(r * 10)), int(round(r * 6))]


def test_mel_filters_match_golden(golden_logmel):
    for n in (80, 128):
        np.testing.assert_allclose(mel_filters(n).numpy(), golden_logmel[f"filters{n}"], atol=2e-7)


def test_lora_structure_and_merge():
    m = _tiny()
    n_lin = sum(isinstance(x, Linear) for x in m.modules())
    assert n_lin == 64  # 64 Linears in whisper-tiny
    lora.apply_lora(m, {"rank": 16, "lora_alpha": 32, "lora_dropout": 0.1})
    names = dict(m.named_parameters())
    assert "decoder.blocks.0.cross_attn.query.parametrizations.weight.0.lora_A" in names
    assert "decoder.blocks.0.cross_attn.query.parametrizations.weight.original" in names
    trainable = [n for n, p in names.items() if p.requires_grad]
    assert trainable and all("lora" in n for n in trainable)
    assert sum(names[n].numel() for n in trainable) == 1081344  # 1.08 M
    q = m.decoder.blocks[0].cross_attn.query
    ad = q.parametrizations.weight[0]
    assert ad.lora_A.shape == (16, 384) and ad.lora_B.shape == (384, 16) and ad.scaling == 2.0
    assert ad.lora_B.norm() == 0 and ad.lora_A.norm() > 0
    assert "decoder.blocks.0.cross_attn.query.parametrizations.weight.0.lora_dropout_mask" in m.state_dict()
    m.eval()
    assert torch.equal(q.weight, q.parametrizations.weight.original)  # B = 0: effective weight unchanged
    with torch.no_grad():
        ad.lora_B.normal_()
    expect = q.parametrizations.weight.original + 2.0 * ad.lora_B @ ad.lora_A
    torch.testing.assert_close(q.weight, expect)
    assert lora.is_lora_enabled(m)
    stats = lora.get_lora_debug_stats(m)
    # the reference's scan (model/lora.py:152-171) reports A of the FIRST adapter and B of the pattern match — checked by
    # running the reference's own get_lora_debug_stats / LoRAUpdateTracker on this model in the build container
    assert stats["param_name"] == "encoder.blocks.0.attn.query.parametrizations.weight.0"
    tr = lora.LoRAUpdateTracker(m)
    assert tr.A_name == "encoder.blocks.0.attn.query.parametrizations.weight.0.lora_A"
    assert tr.B_name == "decoder.blocks.0.cross_attn.query.parametrizations.weight.0.lora_B"
    tr.snapshot()
    assert tr.get_update_norms() == {"delta_A_norm": 0.0, "delta_B_norm": 0.0}
    lora.merge_lora(m)
    assert not lora.is_lora_enabled(m)
    torch.testing.assert_close(m.decoder.blocks[0].cross_attn.query.weight, expect, atol=1e-5, rtol=0)
    assert type(m.decoder.blocks[0].cross_attn.query) is Linear


def test_state_dict_keys_are_openai_whisper_names():
    keys = set(_tiny().state_dict())
    for k in ("encoder.conv1.weight", "encoder.positional_embedding", "encoder.blocks.0.attn.key.weight",
              "encoder.blocks.3.mlp.2.bias", "encoder.ln_post.weight", "decoder.token_embedding.weight",
              "decoder.positional_embedding", "decoder.blocks.0.cross_attn_ln.bias", "decoder.ln.bias"):
        assert k in keys
    assert "encoder.blocks.0.attn.key.bias" not in keys  # K projection has no bias
    assert "decoder.mask" not in keys  # non-persistent


def test_product_path_refuses_cpu_tensors():
    from whisper_finetune.engine import lib as L
    m = _tiny()
    with pytest.raises(L.WftError):
        m(torch.zeros(1, 80, 3000), torch.zeros(1, 4, dtype=torch.long))


def test_any_torch_optimizer_step_invalidates_bf16_shadows():
    """torch's fused optimizers do not bump tensor._version, so the shadow cache keys on an epoch that every
    Optimizer.step() over a shadowed parameter advances (engine/ops.py post-hook)."""
    from whisper_finetune.engine import ops

    p = torch.nn.Parameter(torch.randn(4, 4))
    p.grad = torch.randn(4, 4)
    ops.note_shadowed([p])  # (what LinearGroup.shadows does for the weights it casts)
    for opt in (torch.optim.AdamW([p], lr=1e-3), torch.optim.SGD([p], lr=1e-3)):
        before = ops._SHADOW_EPOCH[0]
        opt.step()
        assert ops._SHADOW_EPOCH[0] == before + 1


def test_lora_mask_pool_draws_once_per_forward_and_is_dropped_on_merge():
    """apply_lora attaches every adapter to ONE mask pool: a forward pre-hook on the model redraws all dropout masks with one
    Bernoulli call, an adapter's mask is its slice ({0, 1/(1-p)} values, [1, in] shape), eval mode and merged models draw nothing."""
    m = _tiny()
    lora.apply_lora(m, {"rank": 4, "lora_alpha": 8, "lora_dropout": 0.5})
    pool = m.__dict__["_wft_lora_pool"]
    ads = [mod.parametrizations.weight[0] for mod in m.modules() if "parametrizations" in mod._modules]
    assert len(pool.adapters) == len(ads) == 64 and pool.total == sum(a.lora_A.shape[1] for a in ads)
    m.train()
    torch.manual_seed(0)
    pool._on_forward(m, ())
    first = [a.draw_mask(True).clone() for a in ads]
    assert all(f.shape == (1, a.lora_A.shape[1]) for f, a in zip(first, ads))
    assert all(set(f.unique().tolist()) <= {0.0, 2.0} for f in first)
    again = [a.draw_mask(True) for a in ads]          # same forward: same masks (a checkpoint recompute sees what the forward saw)
    assert all(torch.equal(x, y) for x, y in zip(first, again))
    pool._on_forward(m, ())                            # next forward: a new draw
    assert any(not torch.equal(x, a.draw_mask(True)) for x, a in zip(first, ads))
    m.eval()
    pool._on_forward(m, ())
    assert all(a.draw_mask(False) is None for a in ads) and pool.buf is None
    m.train(); pool._on_forward(m, ())
    s1, s2 = ads[0].spec(True), ads[0].spec(True)      # one serial number per draw of the pool: the same inside a forward
    assert s1.draw_id == s2.draw_id == pool.serial and s1.key() == s2.key() and s1.owner is ads[0]
    s1_vals, s1_ptr = s1.mask.clone(), s1.mask.data_ptr()
    pool._on_forward(m, ())
    s3 = ads[0].spec(True)                             # ... and a new one with the next forward's masks
    assert s3.draw_id != s1.draw_id and s3.key() != s1.key()
    # the pool's buffer is persistent (s3 views the bytes s1 used to view) and was overwritten in place; s1 is still alive — a
    # forward whose backward has not run — so the draw moved it to a snapshot of ITS values
    assert s3.mask.data_ptr() == s1_ptr and s1.mask.data_ptr() != s1_ptr and torch.equal(s1.mask, s1_vals) and torch.equal(s2.mask, s1_vals)
    del s1, s2
    pool._on_forward(m, ())                            # (s3 alive: moved as well; dead specs cost nothing)
    assert s3.mask.data_ptr() != s1_ptr
    own = lora.LoRAParametrization(8, 8, rank=2, lora_dropout_p=0.5)   # an adapter outside any pool draws its own masks
    assert own.spec(True).draw_id != own.spec(True).draw_id
    lora.merge_lora(m)
    assert "_wft_lora_pool" not in m.__dict__ and not m._forward_pre_hooks and not m._forward_hooks
    assert not m.encoder._forward_pre_hooks and not m.decoder._forward_pre_hooks


def test_deepcopy_of_a_lora_model_owns_its_pool_and_forward_loss():
    """a deep copy of a LoRA model (EMA / teacher / eval-on-a-copy) must evaluate the COPY: its forward_loss shadow, its
    hooks and its adapters all refer to the copy's own pool, the pool's root is the copy, the original is untouched, and merging
    the copy drops the copy's hooks only."""
    import copy

    m = _tiny()
    lora.apply_lora(m, {"rank": 4, "lora_alpha": 8, "lora_dropout": 0.5})
    c = copy.deepcopy(m)
    pm, pc = m.__dict__["_wft_lora_pool"], c.__dict__["_wft_lora_pool"]
    assert pc is not pm and pc.root is c and pm.root is m
    fl = c.__dict__["forward_loss"]
    assert fl.__self__ is pc and fl.__func__ is lora.LoraMaskPool._forward_loss       # a bound method of the copy's pool
    assert m.__dict__["forward_loss"].__self__ is pm
    ads_m = [mod.parametrizations.weight[0] for mod in m.modules() if "parametrizations" in mod._modules]
    ads_c = [mod.parametrizations.weight[0] for mod in c.modules() if "parametrizations" in mod._modules]
    assert all(a._pool is pc for a in ads_c) and all(a._pool is pm for a in ads_m)
    assert [id(a) for a in pc.adapters] == [id(a) for a in ads_c]
    hooks = list(c._forward_pre_hooks.values()) + list(c._forward_hooks.values()) + list(c.encoder._forward_pre_hooks.values())
    assert hooks and all(h.__self__ is pc for h in hooks)
    # a draw on the copy does not touch the original's pool, and the copy's adapters read the copy's masks
    c.train(); m.train()
    serial_m, depth_m = pm.serial, pm.depth
    pc._on_root_forward(c, ())
    assert pc.depth == 1 and pm.depth == depth_m and pm.serial == serial_m and pm.buf is None
    mk = ads_c[3].draw_mask(True)
    assert mk.untyped_storage().data_ptr() == pc.store.untyped_storage().data_ptr() and ads_m[3]._pool.buf is None
    pc._after_root_forward(c, (), None)
    lora.merge_lora(c)
    assert "forward_loss" not in c.__dict__ and not c._forward_pre_hooks and not c.encoder._forward_pre_hooks
    assert len(m._forward_pre_hooks) == 1 and len(m.encoder._forward_pre_hooks) == 1 and "forward_loss" in m.__dict__


def test_grad_fork_sums_like_autograd_in_partial_repeated_and_aborted_passes():
    """ops.GradAccum counted arrivals against every consumer ever registered — a partial backward, two decoder passes with
    different skips or an aborted pass handed the sum over early and lost the tail.  The fork node (ops.grad_fork) counts nothing:
    consumers add into the running sum and return None, autograd runs the fork's backward after the consumers of THIS pass.  A CPU
    stand-in consumer (the real one is LinearFn's backward-data GEMM) against plain autograd."""
    from whisper_finetune.engine import ops

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, accum, fail):
            ctx.save_for_backward(x, w)
            ctx.accum, ctx.fail = accum, fail
            return x * w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            if ctx.fail and ctx.fail.pop():
                raise RuntimeError("boom")
            def dx_fn(run):
                return (g * w).clone() if run is None else run.add_(g * w)
            return ctx.accum.arrive(dx_fn), (g * x).sum(), None, None

    def build(use_fork, fail=None):
        x = torch.arange(1.0, 5.0, requires_grad=True)
        enc = x * 2
        ws = [torch.tensor(float(i + 1), requires_grad=True) for i in range(4)]
        if use_fork:
            xf, acc = ops.grad_fork(enc)
            assert ops.grad_fork(enc)[0] is xf  # cached on the tensor object: every consumer reads the same fork
            outs = [Consumer.apply(xf, w, acc, fail) for w in ws]
        else:
            xf = enc
            outs = [enc * w for w in ws]
        return x, xf, ws, outs

    # 1. every consumer, plus an ordinary autograd user of the forked tensor
    for extra in (False, True):
        ref = build(False)
        got = build(True)
        for x, xf, ws, outs in (ref, got):
            loss = sum(o.sum() for o in outs) + ((xf * 10).sum() if extra else 0)
            loss.backward()
        assert torch.equal(ref[0].grad, got[0].grad) and all(torch.equal(a.grad, b.grad) for a, b in zip(ref[2], got[2]))
    # 2. a partial pass (two of four consumers: stochastic-depth skips), then the other two in a second pass over the same graph
    ref, got = build(False), build(True)
    for x, xf, ws, outs in (ref, got):
        (outs[0].sum() + outs[2].sum()).backward(retain_graph=True)
        first = x.grad.clone()
        (outs[1].sum() + outs[3].sum()).backward()
        x.first = first
    assert torch.equal(ref[0].first, got[0].first) and torch.equal(ref[0].grad, got[0].grad)
    # 3. a pass that never reaches the fork (inputs = one consumer's weight) leaves a sum behind: the next pass drops it
    ref, got = build(False), build(True)
    for x, xf, ws, outs in (ref, got):
        loss = sum(o.sum() for o in outs)
        torch.autograd.grad(loss, [ws[0]], retain_graph=True)
        loss.backward()
    assert torch.equal(ref[0].grad, got[0].grad)
    # 4. an exception in the middle of a pass, then a clean pass
    fail = [False, True, False]  # (popped from the end: the second consumer to run raises)
    x, xf, ws, outs = build(True, fail)
    loss = sum(o.sum() for o in outs)
    try:
        loss.backward(retain_graph=True)
    except RuntimeError as e:
        assert "boom" in str(e)
    else:
        raise AssertionError("expected the injected failure")
    x.grad = None
    for w in ws:
        w.grad = None
    fail.clear()
    loss.backward()
    assert torch.equal(x.grad, torch.full((4,), 2.0 * (1 + 2 + 3 + 4)))


def test_hip_graph_request_is_refused_loudly_where_it_cannot_be_honoured(capsys, monkeypatch):
    """`training.wft_hip_graph: true` (engine/graph.py): train_step captures the micro-batch only for an un-wrapped engine model on
    a HIP device in bf16 mixed precision without host-drawn kernel arguments; anything else says why, once, and runs eagerly."""
    from whisper_finetune.engine import graph as G

    m = _tiny()
    assert "HIP device" in G.why_not(m)
    assert "not an engine Whisper" in G.why_not(torch.nn.Linear(2, 2))
    monkeypatch.setattr(rt, "IS_DISTRIBUTED", False)
    toy = _TinyDDPModel()
    cfg = {**T_CFG, "wft_hip_graph": True}
    for _ in range(2):
        model_utils.train_step(toy, _batches(4), torch.optim.SGD(toy.parameters(), lr=0.1), _Sched(), dict(cfg))
    out = capsys.readouterr().out
    assert out.count("stays on the eager path") == 1 and "bf16 mixed precision" in out
    assert all(p.grad is None for p in toy.parameters())  # the eager path's zero_grad(set_to_none=True) is untouched


def test_optimizer_post_hook_only_counts_optimizers_that_own_shadowed_parameters():
    """engine/ops.py: torch's fused optimizers do not bump tensor._version, so an optimizer step invalidates the bf16 weight
    shadows through a global post-hook — scoped to optimizers that own a parameter the engine has shadowed: an
    EMA / teacher optimizer over other tensors no longer forces a rebuild per step of its own."""
    from whisper_finetune.engine import ops

    mine = torch.nn.Parameter(torch.zeros(4, 4))
    other = torch.nn.Parameter(torch.zeros(4, 4))
    ops.note_shadowed([mine, None])
    opt_mine, opt_other = torch.optim.SGD([mine], lr=0.1), torch.optim.SGD([other], lr=0.1)
    mine.grad, other.grad = torch.ones(4, 4), torch.ones(4, 4)
    e0 = ops._SHADOW_EPOCH[0]
    opt_other.step()
    assert ops._SHADOW_EPOCH[0] == e0
    opt_mine.step()
    assert ops._SHADOW_EPOCH[0] == e0 + 1
    ops.note_shadowed([other])          # shadowed later: the cached classification is refreshed
    opt_other.step()
    assert ops._SHADOW_EPOCH[0] == e0 + 2


def test_flat_train_walk_sets_every_flag_and_defers_to_overriding_modules():
    """Whisper.train(): one flat walk instead of nn.Module.train's recursion (train_step calls it every step); the stock path is
    taken as soon as a sub-module brings its own train()."""
    from whisper_finetune.engine.whisper_model import ModelDimensions

    m = Whisper(ModelDimensions(80, 150, 128, 2, 3, 100, 16, 128, 2, 1))
    mods = list(m.modules())
    assert len(mods) > 50
    assert m.eval() is m and not any(x.training for x in mods)
    assert m.train() is m and all(x.training for x in mods)
    m.train(False)
    assert not any(x.training for x in m.modules())
    with pytest.raises(ValueError):
        m.train("yes")

    calls = []

    class Odd(torch.nn.Module):
        def train(self, mode=True):
            calls.append(mode)
            return super().train(mode)

    m.encoder.blocks[0].add_module("odd", Odd())
    m.train()
    assert calls == [True] and all(x.training for x in m.modules())
    m.eval()
    assert calls == [True, False] and not any(x.training for x in m.modules())
    # the cached bias list of Whisper.forward's hint is dropped with the compute-dtype call the entrypoint makes after module swaps
    m.__dict__["_wft_bias_params"] = []
    m.set_compute_dtype("bf16")
    assert "_wft_bias_params" not in m.__dict__
