"""`get_optimizer(model, optimizer_conf, is_lora_run)` (model/optimizer.py:131-264) for the MI355X build.

adam / adamw map to torch.optim (fused multi-tensor kernels on the GPU) or, with `optimizer.wft: true`, to
`WftAdamW`: one `wft_mt_adamw` launch for the whole parameter list with the clip_grad_norm_ coefficient folded
into the same pass.  `optimizer.muon: true` builds the reference's Muon + auxiliary-Adam parameter groups
(same partition, RMS-matched learning rates and `_lr_group_metadata`) on `WftMuonWithAuxAdam`, whose
Newton-Schulz iterations are batched bf16 MFMA GEMMs.  bitsandbytes 8-bit optimizers raise ImportError as the
reference does when the package is missing."""
from __future__ import annotations

from typing import Dict

import torch


def print_trainable_parameters(model) -> None:
    total = sum(p.numel() for p in model.parameters())
    train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f"trainable params: {train:,} || all params: {total:,} || trainable%: {100 * train / max(total, 1):.4f}")


def _global_sumsq(optimizer, params):
    """device f32 [1] = sum of squares of every gradient (one fixed-order multi-tensor reduction).  The element-count table is
    built once for the optimizer's whole parameter list; a parameter without a gradient this step is a NULL row."""
    from whisper_finetune.engine import kernels as K

    key = tuple(id(p) for p in params)
    ent = optimizer.__dict__.get("_clip_table")
    if ent is None or ent[0] != key:
        ent = optimizer.__dict__["_clip_table"] = (key, K.TensorTable(params))
    return K.mt_sumsq(ent[1], [p.grad for p in params])


class _FusedClipMixin:
    """`fuse_clip_grad_norm(max_norm)`: train_step calls this INSTEAD of torch.nn.utils.clip_grad_norm_
    (model/model_utils.py:107); the next step() then computes the global gradient norm in one multi-tensor pass
    and applies min(1, max_norm / (norm + 1e-6)) inside the update kernels — the gradients are never rewritten."""

    _pending_max_norm = None
    last_grad_norm = None  # device f32 [1] after a fused-clip step (what clip_grad_norm_ would have returned)

    def fuse_clip_grad_norm(self, max_norm: float) -> None:
        if max_norm is None or max_norm <= 0:
            raise ValueError(f"max_norm must be > 0, got {max_norm}")
        self._pending_max_norm = float(max_norm)

    def _take_clip(self):
        max_norm, self._pending_max_norm = self._pending_max_norm, None
        params = [p for g in self.param_groups for p in g["params"]]
        if max_norm is None or not any(p.grad is not None for p in params):
            return None, 0.0
        sumsq = _global_sumsq(self, params)
        self.last_grad_norm = sumsq.sqrt()
        return sumsq, max_norm


def _adamw_groups_step(optimizer, groups, sumsq, max_norm):
    """torch.optim.AdamW semantics for the given param groups, one wft_mt_adamw launch per (group, step count)."""
    from whisper_finetune.engine import kernels as K

    for group in groups:
        b1, b2 = group["betas"]
        by_step = {}
        for p in group["params"]:
            if p.grad is None:
                continue
            st = optimizer.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] += 1
            by_step.setdefault(int(st["step"]), []).append(p)
        for step, ps in by_step.items():
            # A parameter without a gradient this step is SKIPPED (torch.optim.AdamW's rule, which is what the reference runs;
            # the aux-Adam branch of the external muon package would zero-fill it: no moment decay here).  With stochastic depth
            # the set of parameters that have a gradient changes from step to step, so the tables are kept in a small LRU
            # (the common sets recur; an unbounded dict grew by one table — device index arrays included — per new set).
            key = tuple(id(p) for p in ps)
            cache = optimizer.__dict__.setdefault("_tables", {})
            table = cache.pop(key, None)
            if table is None:
                table = K.TensorTable(ps)
                while len(cache) >= 8:
                    cache.pop(next(iter(cache)))
            cache[key] = table  # (re-inserted last: dicts keep insertion order)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            states = [optimizer.state[p] for p in ps]
            K.mt_adamw(table, ps, grads, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states], group["lr"], b1, b2, group["eps"],
                       group["weight_decay"], 1 - b1 ** step, 1 - b2 ** step, sumsq, max_norm)


class WftAdamW(_FusedClipMixin, torch.optim.Optimizer):
    """AdamW (decoupled weight decay, bias correction; torch.optim.AdamW's state names) whose update is ONE libwft
    launch for the whole parameter list (wft_mt_adamw, 28 B/parameter), with clip_grad_norm_ folded in."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built into wft_mt_adamw")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        sumsq, max_norm = self._take_clip()
        _adamw_groups_step(self, self.param_groups, sumsq, max_norm)
        _note_homes(self)
        return loss  # the bf16 weight shadows are invalidated by the global optimizer post-hook (engine/ops.py)


Q8_BLOCK, MIN_8BIT_SIZE = 2048, 4096


def create_dynamic_map(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    """The ascending 256-entry dynamic (tree) quantisation map of bitsandbytes' 8-bit optimizers (`functional.create_dynamic_map`,
    restated from the published scheme — Dettmers et al. 2022; the package is not vendored in the reference: model/optimizer.py:243,253
    import it).  Decade i of 7 (10^-6 .. 10^0) holds 2^i (signed) / 2^(i+1) (unsigned) values, the centres of a linear partition of
    (0.1, 1) * 10^(i-6); both signs for the signed map; plus the codes 0 and 1."""
    data = []
    non_sign_bits = total_bits - 1
    for i in range(max_exponent_bits):
        items = 2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1
        bounds = torch.linspace(0.1, 1, items, dtype=torch.float64)
        means = (bounds[:-1] + bounds[1:]) / 2.0
        scale = 10.0 ** (-(max_exponent_bits - 1) + i)
        data += (scale * means).tolist()
        if signed:
            data += (-scale * means).tolist()
    data += [0.0, 1.0]
    assert len(data) == 2 ** total_bits
    return torch.tensor(sorted(data), dtype=torch.float64).to(torch.float32)


class WftAdamW8bit(_FusedClipMixin, torch.optim.Optimizer):
    """`bnb.optim.AdamW8bit` / `bnb.optim.Adam8bit` (the reference's `optimizer.8bit: True`, model/optimizer.py:241-256) on libwft:
    block-wise dynamic-quantised moments — state1 / state2 one byte per element in blocks of 2 048 with absmax1 / absmax2 per block
    and the two 256-entry maps qmap1 / qmap2 (bitsandbytes' state names) — updated by ONE wft_mt_adamw8 launch for the whole
    parameter list, 16 B per parameter of HBM traffic instead of 28, 2 B per parameter of optimizer state instead of 8.  Tensors
    with fewer than 4 096 elements keep fp32 moments (the package's `min_8bit_size`) and go through wft_mt_adamw.
    PARITY UNPINNED: bitsandbytes is not installed here; the arithmetic follows the published algorithm (oracle/adam8bit_oracle.py:
    decoupled weight decay applied after the update, eps scaled by sqrt(1 - beta2^t) as in the package's kernel)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, min_8bit_size=MIN_8BIT_SIZE):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built into wft_mt_adamw8")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.min_8bit_size = int(min_8bit_size)
        self._qmaps = {}

    def _maps(self, device):
        ent = self._qmaps.get(device)
        if ent is None:
            ent = self._qmaps[device] = (create_dynamic_map(True).to(device), create_dynamic_map(False).to(device))
        return ent

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict casts every state tensor except `step` to the parameter's dtype: the code bytes come
        back as float32 (values 0..255, exact).  Restore uint8 codes, re-share the two maps, keep `step` a Python int (bitsandbytes
        overrides load_state_dict for the same reason)."""
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            for k in ("state1", "state2"):
                if k in st and st[k].dtype != torch.uint8:
                    st[k] = st[k].round().to(torch.uint8).contiguous()
            for k in ("absmax1", "absmax2", "exp_avg", "exp_avg_sq"):
                if k in st and st[k].dtype != torch.float32:
                    st[k] = st[k].float()
            if "state1" in st:
                st["qmap1"], st["qmap2"] = self._maps(p.device)
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())

    @torch.no_grad()
    def step(self, closure=None):
        from whisper_finetune.engine import kernels as K

        loss = closure() if closure is not None else None
        sumsq, max_norm = self._take_clip()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step8, by_step32 = {}, {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    if p.numel() >= self.min_8bit_size:
                        nb = (p.numel() + Q8_BLOCK - 1) // Q8_BLOCK
                        st["state1"] = torch.zeros(p.numel(), dtype=torch.uint8, device=p.device)
                        st["state2"] = torch.zeros(p.numel(), dtype=torch.uint8, device=p.device)
                        st["absmax1"] = torch.zeros(nb, dtype=torch.float32, device=p.device)
                        st["absmax2"] = torch.zeros(nb, dtype=torch.float32, device=p.device)
                        st["qmap1"], st["qmap2"] = self._maps(p.device)
                    else:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                (by_step8 if "state1" in st else by_step32).setdefault(int(st["step"]), []).append(p)
            cache = self.__dict__.setdefault("_tables", {})

            def table_of(ps, tag):
                key = (tag,) + tuple(id(p) for p in ps)
                table = cache.pop(key, None)
                if table is None:
                    table = K.TensorTable(ps)
                    while len(cache) >= 16:
                        cache.pop(next(iter(cache)))
                cache[key] = table
                return table

            for step, ps in by_step8.items():
                sts = [self.state[p] for p in ps]
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                q1, q2 = self._maps(ps[0].device)
                K.mt_adamw8(table_of(ps, 8), ps, grads, [s["state1"] for s in sts], [s["state2"] for s in sts], [s["absmax1"] for s in sts],
                            [s["absmax2"] for s in sts], q1, q2, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                            1 - b1 ** step, 1 - b2 ** step, sumsq, max_norm)
            for step, ps in by_step32.items():
                # fp32 moments for the small tensors, same update rule (decay after the update) expressed through wft_mt_adamw's
                # torch-AdamW form: p (1 - lr wd) - step (...) differs from (p - step (...)) (1 - lr wd) by lr wd step (...) — second
                # order in lr; the package does the latter, and so does this call by running the decay as a separate scale
                sts = [self.state[p] for p in ps]
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                K.mt_adamw(table_of(ps, 32), ps, grads, [s["exp_avg"] for s in sts], [s["exp_avg_sq"] for s in sts], group["lr"], b1, b2,
                           group["eps"], 0.0, 1 - b1 ** step, 1 - b2 ** step, sumsq, max_norm)
                if group["weight_decay"] > 0:
                    torch._foreach_mul_(ps, 1.0 - group["lr"] * group["weight_decay"])
        _note_homes(self)
        return loss


def _note_homes(optimizer) -> None:
    """Where this step's fp32 weight gradients lived (DDP bucket views under gradient_as_bucket_view): the next backward's
    weight-gradient GEMMs write there directly (engine/ops.py `note_grad_homes`)."""
    from whisper_finetune.engine import ops

    # from the SECOND step on: torch DDP rebuilds its buckets at the start of its second iteration, and views of the first
    # iteration's buckets noted here would keep those 6 GB alive beside the new ones for a step (measured: +5.8 GiB peak)
    n = optimizer.__dict__["_wft_steps"] = optimizer.__dict__.get("_wft_steps", 0) + 1
    if n >= 2:
        ops.note_grad_homes(p for g in optimizer.param_groups for p in g["params"])


class WftMuonWithAuxAdam(_FusedClipMixin, torch.optim.Optimizer):
    """`muon.SingleDeviceMuonWithAuxAdam` / `muon.MuonWithAuxAdam` (third-party `muon` package the reference imports at
    model/optimizer.py:171; published algorithm restated in oracle/whisper_oracle.py) on libwft kernels.

    Param groups are the reference's: {"params", "lr", "momentum", "weight_decay", "use_muon": True} and
    {"params", "lr", "betas", "eps", "weight_decay", "use_muon": False}.  Muon parameters of one shape are stepped
    together: momentum/nesterov, bf16 cast, Frobenius normalisation, five Newton-Schulz iterations as BATCHED bf16
    MFMA GEMMs, update.  In a multi-process job (torch.distributed initialised, world > 1: what selects
    `muon.MuonWithAuxAdam` in the reference, model/optimizer.py:227-228) the Newton-Schulz work is SHARDED: every
    same-shape bucket is dealt to the ranks in contiguous chunks, a rank keeps momentum and orthogonalises only its own
    matrices, the bf16 updates are all-gathered over RCCL and every rank applies all of them in fp32 — parameters stay
    bit-identical across ranks and 8 ranks no longer repeat 8x the same GEMMs (the package gathers the
    updated fp32 parameters instead: twice the bytes for the same result).  The auxiliary-Adam groups are element-wise
    and stay replicated."""

    def __init__(self, param_groups):
        for group in param_groups:
            assert "use_muon" in group
            if group["use_muon"]:
                group["lr"] = group.get("lr", 0.02)
                group["momentum"] = group.get("momentum", 0.95)
                group["weight_decay"] = group.get("weight_decay", 0)
                assert set(group.keys()) == {"params", "lr", "momentum", "weight_decay", "use_muon"}
            else:
                group["lr"] = group.get("lr", 3e-4)
                group["betas"] = group.get("betas", (0.9, 0.95))
                group["eps"] = group.get("eps", 1e-10)
                group["weight_decay"] = group.get("weight_decay", 0)
                assert set(group.keys()) == {"params", "lr", "betas", "eps", "weight_decay", "use_muon"}
        super().__init__(param_groups, dict())

    def _muon_buckets(self, gi, group, shard):
        """The group's bucket plan, rebuilt when the parameter list, the sharding, a parameter's storage or a momentum buffer
        (optimizer.load_state_dict replaces those) is not what the plan was built from: 2 x 1 024 address compares per step
        instead of 3 x 1 024 fresh views and their checks — the host side of a Muon step used to leave the GPU idle between
        buckets (profiles/r03_lora_gap_analysis.log)."""
        plans = self.__dict__.setdefault("_muon_plans", {})
        key = (tuple(id(p) for p in group["params"]), None if shard is None else (shard[0], shard[1]))
        ent = plans.get(gi)
        if ent is not None and ent[0] == key:
            ok = all(v.data_ptr() == p.data_ptr() for _, _, ps, _, pv, _ in ent[1] for p, v in zip(ps, pv)) and \
                all(self.state[p].get("momentum_buffer") is not None and v.data_ptr() == self.state[p]["momentum_buffer"].data_ptr()
                    for _, _, _, own, _, bv in ent[1] for p, v in zip(own, bv))
            if ok:
                return ent[1]
        ent = plans[gi] = (key, _muon_buckets_build(self, group, shard))
        return ent[1]

    @torch.no_grad()
    def step(self, closure=None):
        from whisper_finetune.engine import kernels as K

        loss = closure() if closure is not None else None
        # (muon.py gives a Muon parameter without a gradient a zero one, "force synchronization": its momentum decays and the
        # update comes from the momentum alone.  Here such a parameter's table row is NULL and wft_muon_momentum_mt computes
        # exactly that without the 120 allocations + fills per step a stochastic-depth run made of it; zeros add nothing to
        # the clipping norm either)
        sumsq, max_norm = self._take_clip()
        shard = _muon_shard()
        for gi, group in enumerate(self.param_groups):
            if not group["use_muon"]:
                continue
            for rows, cols, ps, own, pviews, bviews in self._muon_buckets(gi, group, shard):
                grads = []
                for p in own:
                    g = p.grad
                    if g is not None and (g.ndim != 2 or not g.is_contiguous()):
                        g = (g if g.is_contiguous() else g.contiguous()).view(rows, cols)
                    grads.append(g)
                K.muon_group_step(pviews, grads, bviews, group["lr"], group["weight_decay"], group["momentum"], sumsq=sumsq,
                                  max_norm=max_norm, shard=shard, validated=True)
        _adamw_groups_step(self, [g for g in self.param_groups if not g["use_muon"]], sumsq, max_norm)
        _note_homes(self)
        return loss


def _muon_buckets_build(optimizer, group, shard):
    """Same-shape buckets of one Muon group with everything that does not change from step to step: the [rows, cols] views of
    the parameters and of the owned momentum buffers (created here, on the owning rank only), checked once."""
    buckets = {}
    for p in group["params"]:
        if p.ndim < 2:
            raise ValueError("Muon parameters must have ndim >= 2")
        buckets.setdefault((p.shape[0], p[0].numel()), []).append(p)
    out = []
    for (rows, cols), ps in buckets.items():
        if shard is None:
            own = ps
        else:
            per = (len(ps) + shard[1] - 1) // shard[1]
            own = ps[shard[0] * per:(shard[0] + 1) * per]
        for p in own:  # momentum lives on the owning rank only
            if not optimizer.state[p]:
                optimizer.state[p]["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        pviews = [p.data.view(rows, cols) for p in ps]
        bviews = [optimizer.state[p]["momentum_buffer"].view(rows, cols) for p in own]
        for t in pviews + bviews:
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("WftMuonWithAuxAdam needs contiguous f32 parameters on a HIP device (there is no CPU path)")
        out.append((rows, cols, ps, own, pviews, bviews))
    return out


def _muon_shard():
    """(rank, world, all_gather) when the Newton-Schulz work is sharded over the process group, else None."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    world = dist.get_world_size()

    def all_gather(out, inp):
        # one all-gather per same-shape bucket, the into-tensor form on EVERY backend that has it (RCCL in production; gloo
        # takes it for CPU tensors, so the world-size-2 CPU test runs the production call with its padded `inp` and
        # world * per sized `out`); gloo with device tensors falls back to the list form over the same layout
        try:
            dist.all_gather_into_tensor(out, inp)
        except (RuntimeError, NotImplementedError):
            if dist.get_backend() == "nccl":
                raise
            dist.all_gather(list(out.chunk(world, dim=0)), inp)

    return dist.get_rank(), world, all_gather


def _partition_muon_params(model, ndim_threshold: int = 2):
    """Muon gets the >= ndim_threshold-D parameters INSIDE encoder.blocks / decoder.blocks; everything else (gains,
    biases, embeddings, conv stem, final norms) stays on the auxiliary Adam (model/optimizer.py:9-52)."""
    block_ids = {id(p) for blk in list(model.encoder.blocks) + list(model.decoder.blocks) for p in blk.parameters()}
    muon, aux, seen = [], [], set()
    for _, p in model.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        (muon if id(p) in block_ids and p.ndim >= ndim_threshold else aux).append(p)
        seen.add(id(p))
    expected = {id(p) for p in model.parameters() if p.requires_grad}
    if seen != expected:
        raise RuntimeError(f"Muon parameter partition mismatch: missing={len(expected - seen)}, extra={len(seen - expected)}. "
                           "This should not happen.")
    return muon, aux


def _use_muon_optimizer(optimizer_conf: Dict) -> bool:
    if "muon" in optimizer_conf:
        return bool(optimizer_conf["muon"])
    return optimizer_conf.get("type") == "muon"


def _muon_update_rms_match_scale(param, factor: float = 0.2) -> float:
    """lr multiplier that turns the package's sqrt(max(1, A/B)) update scaling into the paper's 0.2 sqrt(max(A, B)):
    factor * sqrt(B_effective), B_effective = last dim after Muon's flattening (model/optimizer.py:61-85)."""
    if param.ndim < 2:
        raise ValueError("Muon RMS matching requires parameters with ndim >= 2.")
    b_eff = param[0].numel() if param.ndim == 4 else param.shape[-1]
    return float(factor) * (float(b_eff) ** 0.5)


def _build_muon_param_groups(muon_params, base_lr, base_weight_decay, momentum, match_adamw_update_rms, match_factor):
    if not match_adamw_update_rms:
        return [{"params": muon_params, "use_muon": True, "lr": base_lr, "momentum": momentum,
                 "weight_decay": base_weight_decay}]
    grouped = {}
    for p in muon_params:
        scale = _muon_update_rms_match_scale(p, factor=match_factor)
        if scale <= 0:
            raise ValueError(f"Muon RMS match scale must be > 0, got {scale}")
        key = (p.ndim, p[0].numel() if p.ndim == 4 else p.shape[-1])
        if key not in grouped:
            grouped[key] = {"params": [], "use_muon": True, "lr": base_lr * scale, "momentum": momentum,
                            "weight_decay": base_weight_decay / scale if base_weight_decay != 0 else 0.0}
        grouped[key]["params"].append(p)
    return list(grouped.values())


def _get_muon_optimizer(model, optimizer_conf: Dict):
    """model/optimizer.py:163-237: group construction, warnings and `_lr_group_metadata` identical; the optimizer class is
    the libwft one on HIP tensors (and only there: there is no CPU Muon in this build)."""
    if optimizer_conf.get("type") not in (None, "adamw", "muon"):
        print("WARNING: optimizer.type is ignored when optimizer.muon=True. Using MuonWithAuxAdam.")
    if optimizer_conf.get("8bit", False):
        print("WARNING: optimizer.8bit=True is ignored for Muon.")
    ndim_threshold = int(optimizer_conf.get("muon_ndim_threshold", 2))
    if ndim_threshold < 1:
        raise ValueError(f"optimizer.muon_ndim_threshold must be >= 1, got {ndim_threshold}")
    muon_params, aux_params = _partition_muon_params(model, ndim_threshold=ndim_threshold)
    muon_conf = optimizer_conf.get("muon_params", {})
    adamw_conf = optimizer_conf.get("params", {})
    adamw_lr = adamw_conf.get("lr", 3e-4)
    adamw_betas = tuple(adamw_conf.get("betas", (0.9, 0.95)))
    adamw_eps = adamw_conf.get("eps", 1e-10)
    adamw_wd = adamw_conf.get("weight_decay", 0.0)
    match = bool(optimizer_conf.get("muon_match_adamw_update_rms", True))
    factor = float(optimizer_conf.get("muon_match_factor", 0.2))
    if factor <= 0:
        raise ValueError(f"optimizer.muon_match_factor must be > 0, got {factor}")
    if "amsgrad" in adamw_conf:
        print("WARNING: optimizer.params.amsgrad is not used by Muon auxiliary AdamW.")
    muon_lr = muon_conf.get("lr", 0.02)
    muon_momentum = muon_conf.get("momentum", 0.95)
    muon_wd = muon_conf.get("weight_decay", adamw_wd)
    groups = _build_muon_param_groups(muon_params, muon_lr, muon_wd, muon_momentum, match, factor)
    metadata = [{"lr_log_label": "muon", "base_lr_unscaled": muon_lr} for _ in groups]
    if aux_params:
        groups.append({"params": aux_params, "use_muon": False, "lr": adamw_lr, "betas": adamw_betas, "eps": adamw_eps,
                       "weight_decay": adamw_wd})
        metadata.append({"lr_log_label": "aux_adamw", "base_lr_unscaled": adamw_lr})
    if match:
        print(f"Muon RMS matching active: factor={factor}, shared base_lr={muon_lr}, shared weight_decay={muon_wd}")
    print(f"Using WftMuonWithAuxAdam with {len(muon_params)} Muon params and {len(aux_params)} AuxAdamW params")
    optimizer = WftMuonWithAuxAdam(groups)
    optimizer._lr_group_metadata = metadata
    return optimizer


def get_optimizer(model, optimizer_conf: Dict, is_lora_run: bool = False):
    params = [p for p in model.parameters() if p.requires_grad]
    print("---OPTIMIZER----")
    print_trainable_parameters(model)
    if optimizer_conf.get("8bit", False) and is_lora_run:
        print("WARNING: Using 8-bit optimizer with LoRA training.")
    if _use_muon_optimizer(optimizer_conf):
        return _get_muon_optimizer(model, optimizer_conf)
    kind = optimizer_conf["type"]
    kw = dict(optimizer_conf.get("params", {}))
    if "betas" in kw:
        kw["betas"] = tuple(kw["betas"])
    on_gpu = bool(params) and params[0].is_cuda
    if optimizer_conf.get("8bit", False):
        # bnb.optim.Adam8bit / AdamW8bit (model/optimizer.py:241-256).  bitsandbytes has no gfx950 build in this image; the
        # block-wise 8-bit state update is a libwft kernel (HIP tensors only: on the CPU the reference's ImportError stands)
        if kind not in ("adam", "adamw"):
            raise ValueError(f"Unknown optimizer type: {kind}. Must be adam or adamw.")
        if not on_gpu:
            raise ImportError("For using Adam 8bit optimizer you need to have bitsandbytes installed. "
                              "(the libwft 8-bit AdamW runs on HIP tensors only)")
        kw.pop("amsgrad", None)
        if kind == "adam":
            kw.setdefault("weight_decay", 0.0)  # bnb.optim.Adam8bit's default (AdamW8bit: 1e-2); same kernel, decoupled decay
        return WftAdamW8bit(params, **kw)
    # AdamW on HIP tensors runs in libwft (one launch per step, clip folded in) unless optimizer.wft: false asks for torch's
    if kind == "adamw" and optimizer_conf.get("wft", on_gpu) and not kw.get("amsgrad", False):
        kw.pop("amsgrad", None)
        return WftAdamW(params, **kw)
    if kind == "adam":
        return torch.optim.Adam(params, **kw, **({"fused": True} if on_gpu else {}))
    if kind == "adamw":
        return torch.optim.AdamW(params, **kw, **({"fused": True} if on_gpu else {}))
    raise ValueError(f"Unknown optimizer type: {kind}. Must be adam or adamw.")
