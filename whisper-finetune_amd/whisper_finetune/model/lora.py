"""LoRA for the MI355X build: same entry points as the reference's `model/lora.py`
(apply_lora :30-71, disable_all_but_parametrized_grads :14-27, remove_lora/merge_lora :74-89,
print_lora_info, is_lora_enabled, get_lora_debug_stats, LoRAUpdateTracker,
log_lora_debug_info), same parameter / state-dict names as minLoRA's weight parametrization
(`<linear>.parametrizations.weight.original`, `.0.lora_A`, `.0.lora_B`,
`.0.lora_dropout_mask` — SURVEY.md App. A.3).  engine/ops.LinearFn evaluates

    y = x (W + (alpha/r) B (A * m))^T + b

with the effective weight written straight into the bf16 GEMM shadow by `wft_lora_merge` (one HBM-bound pass per
Linear per forward, instead of an fp32 delta + add + cast), so forward and backward-data are the plain GEMMs; the
adapter gradients stay low-rank (dA = (dy sB)^T x, dB = s dy^T (x (A*m)^T): four rank-r GEMMs) instead of the full
d_out x d_in weight-gradient GEMM the parametrization form pays (SURVEY.md finding 7).  `layer.weight` still returns
the effective weight for code that reads it (merge / save / debugging).
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
from torch import nn

from whisper_finetune.engine import ops
from whisper_finetune.engine.whisper_model import Linear as WLinear


class LoRAParametrization(nn.Module):
    """minLoRA's LoRAParametrization.from_linear: A = kaiming_uniform_(a=sqrt(5)) [r, in], B = zeros
    [out, r], scaling = alpha / rank, dropout applied to a [1, in] ones mask (so it drops whole
    input COLUMNS of A, shared by every token of the batch)."""

    def __init__(self, fan_in: int, fan_out: int, rank: int = 4, lora_dropout_p: float = 0.0, lora_alpha: float = 1.0,
                 device=None):
        super().__init__()
        self.lora_A = nn.Parameter(torch.zeros(rank, fan_in, device=device))
        self.lora_B = nn.Parameter(torch.zeros(fan_out, rank, device=device))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        self.lora_alpha, self.rank = lora_alpha, rank
        self.scaling = lora_alpha / rank
        self.lora_dropout_p = float(lora_dropout_p)
        self.register_buffer("lora_dropout_mask", torch.ones(1, fan_in, dtype=self.lora_A.dtype, device=device))
        self.enabled = True

    _pool = None  # LoraMaskPool of the model this adapter belongs to (apply_lora), or None

    def draw_mask(self, training: bool) -> Optional[torch.Tensor]:
        """Dropout(p)(ones[1, in]): a fresh mask per forward of the Linear.  Adapters attached by
        apply_lora take their mask from the model's pool — ONE Bernoulli launch per model forward for all (512) adapters
        instead of one per Linear — anything else draws its own."""
        if self.lora_dropout_p > 0.0 and training:
            if self._pool is not None:
                m = self._pool.mask_of(self)
                if m is not None:
                    return m
            return torch.nn.functional.dropout(self.lora_dropout_mask, self.lora_dropout_p, True)
        return None

    def spec(self, training: bool) -> Optional[ops.LoraSpec]:
        if not self.enabled:
            return None
        m = self.draw_mask(training)
        pool = self._pool
        if m is not None and pool is not None and pool.buf is not None and m.untyped_storage().data_ptr() == pool.buf.untyped_storage().data_ptr():
            # the pool's draw for this forward of the model, under the pool's serial number; the pool keeps track of the specs
            # that view its buffer so that the NEXT draw can move the ones still alive (a backward yet to run) to a snapshot
            sp = ops.LoraSpec(self.lora_A, self.lora_B, self.scaling, m, draw_id=pool.serial, owner=self)
            pool.handed.append(weakref.ref(sp))
            return sp
        return ops.LoraSpec(self.lora_A, self.lora_B, self.scaling, m, owner=self)

    def forward(self, W: torch.Tensor) -> torch.Tensor:
        """Effective weight (off the hot path: merge / inspection)."""
        if not self.enabled:
            return W
        m = self.draw_mask(self.training)
        if W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and not torch.is_grad_enabled():
            from whisper_finetune.engine import kernels as K

            out = torch.empty_like(W)  # native merge kernel W + s*B@(A*mask) (merge_lora, save paths)
            K.lora_merge(W, self.lora_B, self.lora_A, m, self.scaling, out_f32=out)
            return out
        A = self.lora_A if m is None else self.lora_A * m
        return W + (self.lora_B @ A).view(W.shape) * self.scaling


class LoraMaskPool:
    """All dropout masks of a model's adapters as slices of one buffer, redrawn once per forward of the root module (a
    forward pre-hook): `bernoulli_(1 - p) / (1 - p)` over the concatenated input widths — the distribution of minLoRA's
    per-adapter `Dropout(p)(ones[1, in])`, 2 launches instead of 512.  A checkpoint recompute inside the same forward sees the
    same masks (they change only when the root forward starts).

    The buffer is persistent (the batched refresh kernel's table holds addresses inside it) and overwritten in place by the next
    draw, so a draw first looks for LoraSpecs of the PREVIOUS draw that are still alive — a forward whose backward has not run
    yet: `loss = model(a) + model(b)`, R-Drop, a train-mode probe — and moves their masks to a snapshot of the old values; their
    backward then rebuilds its operands from exactly the masks its forward used (the per-Linear path: the serial differs).  In the
    ordinary loop (forward, backward, step) no spec survives to the next draw and nothing is copied.
    Train-mode calls that enter below the root (`model.encoder(x)`) reach the same draw through hooks on the root's direct
    children: they draw when no forward of the root is in progress; `model.forward_loss(...)` is wrapped to count as one forward
    of the root."""

    def __init__(self, root: nn.Module):
        self.adapters = []
        self.offsets = {}
        self.buf = None      # this forward's masks, or None (eval, p = 0, mixed rates)
        self.store = None    # the persistent buffer `buf` points at (wft_lora_refresh_mt's table holds addresses inside it)
        self.serial = 0      # ops.LoraSpec draw id of the current masks
        self.total = 0
        self.handed = []     # weak references to the LoraSpecs that view `store` under the current serial
        self.depth = 0       # > 0 while a forward of the root is running
        self.handle = root.register_forward_pre_hook(self._on_root_forward)
        self.handles = [root.register_forward_hook(self._after_root_forward, always_call=True)]
        for child in root.children():  # (encoder, decoder): entry points of calls that bypass root.__call__
            self.handles.append(child.register_forward_pre_hook(self._on_child_forward))
        self.root = root
        # `model.forward_loss(...)` is a plain method, not a child module: shadowed by a BOUND METHOD of the pool so that it counts as
        # ONE forward of the root (one draw; without this its encoder and decoder calls each looked like an entry below the root — a
        # second draw while the first one's specs were alive, and with checkpointed blocks a recompute against the wrong masks).
        # A bound method, not a closure: copy.deepcopy re-binds it to the copy's pool (a closure is copied by reference and would
        # evaluate the ORIGINAL model from the copy — EMA / teacher / eval-on-a-copy flows).
        if callable(getattr(type(root), "forward_loss", None)) and "forward_loss" not in root.__dict__:
            root.__dict__["forward_loss"] = self._forward_loss
        # all merged shadows / gradient-GEMM operands of the model in one launch per training forward (WFT_LORA_BATCH=0: the
        # per-Linear kernels, A/B runs)
        self.plan = ops.LoraRefreshPlan() if os.environ.get("WFT_LORA_BATCH", "1") != "0" else None

    def _forward_loss(self, *args, **kwargs):
        root = self.root
        inner = type(root).forward_loss
        if self.depth > 0:  # reached through root.forward(..., targets=...): that call already is the forward of the model
            return inner(root, *args, **kwargs)
        self._on_root_forward(root, args)
        try:
            return inner(root, *args, **kwargs)
        finally:
            self._after_root_forward(root, args, None)

    def __deepcopy__(self, memo):
        """The pool of a deep-copied model: the copy's root, adapters and hook handles (the hooks themselves are bound methods of
        the pool inside the root's hook dictionaries: deepcopy re-binds them to this object through the memo); offsets re-keyed to
        the copied adapters; no masks drawn yet and an empty refresh plan — the original's plan holds device addresses of the
        original's shadows, the copy rebuilds its own on its first training forward."""
        import copy

        new = object.__new__(type(self))
        memo[id(self)] = new
        new.root = copy.deepcopy(self.root, memo)
        new.adapters = [copy.deepcopy(a, memo) for a in self.adapters]
        new.offsets = {id(b): self.offsets[id(a)] for a, b in zip(self.adapters, new.adapters)}
        new.buf = new.store = None
        new.serial = 0
        new.total = self.total
        new.handed = []
        new.depth = 0
        new.handle = copy.deepcopy(self.handle, memo)
        new.handles = [copy.deepcopy(h, memo) for h in self.handles]
        new.plan = ops.LoraRefreshPlan() if self.plan is not None else None
        return new

    def add(self, adapter: "LoRAParametrization") -> None:
        self.offsets[id(adapter)] = (self.total, adapter.lora_A.shape[1])
        self.total += adapter.lora_A.shape[1]
        self.adapters.append(adapter)
        adapter.__dict__["_pool"] = self

    def _on_root_forward(self, module, inputs):
        self.depth += 1
        self._on_forward(module, inputs)

    def _after_root_forward(self, module, inputs, output):
        self.depth = max(0, self.depth - 1)

    def _on_child_forward(self, module, inputs):
        if self.depth == 0:  # entered below the root: this call is its own "forward of the model"
            self._on_forward(self.root, inputs)

    def _detach_live_specs(self) -> None:
        """Specs of the previous draw whose backward is still to come keep the OLD mask values (a snapshot)."""
        live = [sp for sp in (r() for r in self.handed) if sp is not None and sp.mask is not None]
        self.handed = []
        if live and self.store is not None:
            snap = self.store.clone()
            for sp in live:
                off, n = self.offsets[id(sp.owner)]
                sp.mask = snap[off:off + n].view(1, n)

    def _on_forward(self, module, inputs):
        if not module.training or not self.adapters:
            self.buf = None
            return
        p = self.adapters[0].lora_dropout_p
        if any(a.lora_dropout_p != p for a in self.adapters):
            self.buf = None  # mixed rates: every adapter draws its own
            return
        if p > 0.0:
            A0 = self.adapters[0].lora_A
            if self.store is None or self.store.device != A0.device or self.store.dtype != A0.dtype or self.store.numel() != self.total:
                self.store = torch.empty(self.total, device=A0.device, dtype=A0.dtype)
                if self.plan is not None:
                    self.plan.dirty = True
            self._detach_live_specs()
            self.buf = self.store.bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
            self.serial = next(ops.LoraSpec._draws)
        else:
            self.buf = None
        if self.plan is not None and self.adapters[0].lora_A.is_cuda and getattr(module, "compute_dtype", "bf16") != "fp32":
            self.plan.refresh(self.mask_of, self.serial)

    def mask_of(self, adapter) -> Optional[torch.Tensor]:
        if self.buf is None:
            return None
        off, n = self.offsets[id(adapter)]
        return self.buf[off:off + n].view(1, n)


class ParametrizationList(nn.ModuleList):
    """Holds `original` (the base weight) and the adapter at index 0 — the key layout
    torch.nn.utils.parametrize produces and the reference's checkpoints carry."""

    def __init__(self, original: nn.Parameter, adapter: LoRAParametrization):
        super().__init__([adapter])
        self.original = original


def _effective_weight(self):
    plist = self.parametrizations.weight
    return plist[0](plist.original)


def add_lora_to_linear(layer: WLinear, rank: int, lora_alpha: float, lora_dropout_p: float) -> None:
    if "parametrizations" in layer._modules:
        return
    W = layer._parameters.pop("weight")
    adapter = LoRAParametrization(W.shape[1], W.shape[0], rank=rank, lora_dropout_p=lora_dropout_p,
                                  lora_alpha=lora_alpha, device=W.device)
    layer.parametrizations = nn.ModuleDict({"weight": ParametrizationList(W, adapter)})
    cls = layer.__class__
    layer.__class__ = type(f"Parametrized{cls.__name__}", (cls,), {"weight": property(_effective_weight), "_wft_base_cls": cls})


def disable_all_but_parametrized_grads(model: nn.Module) -> None:
    """Freeze every parameter whose name does not contain "lora" (model/lora.py:14-27)."""
    for name, p in model.named_parameters():
        if "lora" not in name.lower():
            p.requires_grad = False


def apply_lora(model: nn.Module, lora_config: dict, train_only_decoder: bool = False, train_only_encoder: bool = False) -> None:
    """Attach adapters to every whisper Linear (q, k, v, out, mlp.0, mlp.2 — not conv / embedding) of the
    model, or of the decoder / encoder only, then freeze the rest (model/lora.py:30-71).
    lora_config keys: rank, lora_alpha, lora_dropout."""
    cfg = dict(lora_config)
    p_drop = cfg.pop("lora_dropout", cfg.pop("lora_dropout_p", 0.0))
    rank = int(cfg.pop("rank", 4))
    alpha = cfg.pop("lora_alpha", 1)
    root = model.decoder if train_only_decoder else model.encoder if train_only_encoder else model
    pool = model.__dict__.get("_wft_lora_pool")
    if pool is None:
        pool = model.__dict__["_wft_lora_pool"] = LoraMaskPool(model)
    for m in root.modules():
        if isinstance(m, WLinear) and "parametrizations" not in m._modules:
            add_lora_to_linear(m, rank, alpha, p_drop)
            pool.add(m.parametrizations.weight[0])
    disable_all_but_parametrized_grads(model)


def _strip(layer: nn.Module, merge: bool) -> None:
    if "parametrizations" not in layer._modules:
        return
    plist = layer.parametrizations["weight"] if "weight" in layer.parametrizations else None
    if not isinstance(plist, ParametrizationList):
        # a layer parametrized through torch.nn.utils.parametrize (minLoRA on a plain nn.Linear): the reference's route
        import torch.nn.utils.parametrize as parametrize

        for attr in list(layer.parametrizations.keys()):
            parametrize.remove_parametrizations(layer, attr, leave_parametrized=merge)
        return
    with torch.no_grad():
        W = plist[0].eval()(plist.original) if merge else plist.original
        new = nn.Parameter(W.detach().clone(), requires_grad=plist.original.requires_grad)
    del layer._modules["parametrizations"]
    layer.__class__ = layer.__class__._wft_base_cls
    layer._parameters["weight"] = new
    layer.__dict__.pop("_wft_group", None)


def _drop_pool(model: nn.Module) -> None:
    pool = model.__dict__.pop("_wft_lora_pool", None)
    if pool is not None:
        pool.handle.remove()
        for h in pool.handles:
            h.remove()
        model.__dict__.pop("forward_loss", None)


def remove_lora(model: nn.Module) -> None:
    """Drop the adapters and restore the original base weights (model/lora.py:74-80)."""
    model.apply(lambda m: _strip(m, merge=False))
    _drop_pool(model)


def merge_lora(model: nn.Module) -> None:
    """Fold W + s*B@A into the base weights (model/lora.py:83-89)."""
    model.apply(lambda m: _strip(m, merge=True))
    _drop_pool(model)


def print_lora_info(model: nn.Module) -> None:
    total = sum(p.numel() for p in model.parameters())
    train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f"LoRA trainable parameters: {train:,}")
    print(f"Total parameters: {total:,}")
    print(f"Trainable %: {100 * train / total:.4f}%")


def is_lora_enabled(model: nn.Module) -> bool:
    return any("lora" in n.lower() for n, _ in model.named_parameters())


_REP = "decoder.blocks.0.cross_attn.query.parametrizations.weight"


def _representative(model: nn.Module, pattern: str):
    """The (name, parameter) pairs the reference reports on (model/lora.py:152-171,207-216).  Its scan takes the FIRST
    lora_A / lora_B it meets while nothing has been found yet, and afterwards only names containing `pattern` — so with the
    default pattern A comes from the first adapted Linear of the model and B from `decoder.blocks.0.cross_attn.query`.
    Reproduced as is: the logged numbers are compared across runs of the two code bases."""
    a = b = None
    for n, p in model.named_parameters():
        if pattern in n or (a is None and b is None):
            if "lora_A" in n and a is None:
                a = (n, p)
            elif "lora_B" in n and b is None:
                b = (n, p)
        if a is not None and b is not None:
            break
    return a, b


def get_lora_debug_stats(model: nn.Module, representative_module_pattern: str = _REP) -> dict:
    stats = dict.fromkeys(["lora_A_norm", "lora_B_norm", "lora_A_grad_norm", "lora_B_grad_norm",
                           "lora_A_grad_abs_max", "lora_B_grad_abs_max", "param_name"])
    a, b = _representative(model, representative_module_pattern)
    for tag, item in (("A", a), ("B", b)):
        if item is None:
            continue
        n, p = item
        if tag == "A":
            stats["param_name"] = n.replace(".lora_A", "")
        stats[f"lora_{tag}_norm"] = p.detach().float().norm().item()
        if p.grad is not None:
            g = p.grad.detach().float()
            stats[f"lora_{tag}_grad_norm"] = g.norm().item()
            stats[f"lora_{tag}_grad_abs_max"] = g.abs().max().item()
    return stats


class LoRAUpdateTracker:
    """||dA||, ||dB|| of one representative adapter across optimizer steps (model/lora.py:180-252)."""

    def __init__(self, model: nn.Module, representative_module_pattern: str = _REP):
        self.model, self.pattern = model, representative_module_pattern
        a, b = _representative(model, representative_module_pattern)
        self.A_name, self._A = a if a else (None, None)
        self.B_name, self._B = b if b else (None, None)
        self.prev_A = self.prev_B = None

    def snapshot(self):
        if self._A is not None:
            self.prev_A = self._A.detach().clone().float()
        if self._B is not None:
            self.prev_B = self._B.detach().clone().float()

    def get_update_norms(self) -> dict:
        out = {"delta_A_norm": None, "delta_B_norm": None}
        if self._A is not None and self.prev_A is not None:
            out["delta_A_norm"] = (self._A.detach().float() - self.prev_A).norm().item()
        if self._B is not None and self.prev_B is not None:
            out["delta_B_norm"] = (self._B.detach().float() - self.prev_B).norm().item()
        return out


def log_lora_debug_info(model: nn.Module, step: int, tracker: Optional[LoRAUpdateTracker] = None, log_to_wandb: bool = True) -> dict:
    """Collect the adapter norms / grad norms / last update norms and log them on rank 0
    (called from train_step at eval steps, model/model_utils.py:95-105)."""
    import whisper_finetune.runtime as rt

    stats = get_lora_debug_stats(model)
    if tracker is not None:
        stats.update(tracker.get_update_norms())
    if log_to_wandb:
        payload = {f"lora_debug/{k}": v for k, v in stats.items() if v is not None and k != "param_name"}
        if payload:
            rt.log(payload, step=step)
    return stats
