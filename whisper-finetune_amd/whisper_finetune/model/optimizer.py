"""`get_optimizer(model, optimizer_conf, is_lora_run)` (model/optimizer.py:131-264) for the MI355X build.

adam / adamw map to torch.optim (fused multi-tensor kernels on the GPU) or, with `optimizer.wft: true`, to
`WftAdamW`, which steps every parameter with the `wft_adamw_step` kernel and folds the clip_grad_norm_
coefficient into the same pass.  bitsandbytes 8-bit optimizers and Muon are SURVEY.md §8f-1 ("next") and
raise a clear error here rather than silently substituting another optimizer."""
from __future__ import annotations

from typing import Dict

import torch


def print_trainable_parameters(model) -> None:
    total = sum(p.numel() for p in model.parameters())
    train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f"trainable params: {train:,} || all params: {total:,} || trainable%: {100 * train / max(total, 1):.4f}")


class WftAdamW(torch.optim.Optimizer):
    """AdamW (decoupled weight decay, bias correction) whose update runs in libwft.  `grad_scale` (a device
    scalar, e.g. the clipping coefficient) is applied to the gradient inside the kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built into wft_adamw_step")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = None

    @torch.no_grad()
    def step(self, closure=None):
        from whisper_finetune.engine import kernels as K
        from whisper_finetune.engine import ops

        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                K.adamw_step(p, g, st["m"], st["v"], None, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                             1 - b1 ** st["step"], 1 - b2 ** st["step"], self.grad_scale)
        ops.bump_shadow_epoch()  # parameters changed through raw pointers: invalidate the bf16 shadows
        return loss


def get_optimizer(model, optimizer_conf: Dict, is_lora_run: bool = False):
    params = [p for p in model.parameters() if p.requires_grad]
    print("---OPTIMIZER----")
    print_trainable_parameters(model)
    if optimizer_conf.get("8bit", False):
        raise ImportError("8-bit optimizers need bitsandbytes, which has no gfx950 build in this environment "
                          "(set optimizer.8bit: False)")
    if optimizer_conf.get("muon", False) or optimizer_conf.get("type") == "muon":
        raise NotImplementedError("Muon + AuxAdam is not built yet (SURVEY.md §8f-1); use optimizer.type: adamw")
    kind = optimizer_conf["type"]
    kw = dict(optimizer_conf.get("params", {}))
    if "betas" in kw:
        kw["betas"] = tuple(kw["betas"])
    on_gpu = bool(params) and params[0].is_cuda
    if kind == "adamw" and optimizer_conf.get("wft", False):
        return WftAdamW(params, **kw)
    if kind == "adam":
        return torch.optim.Adam(params, **kw, **({"fused": True} if on_gpu else {}))
    if kind == "adamw":
        return torch.optim.AdamW(params, **kw, **({"fused": True} if on_gpu else {}))
    raise ValueError(f"Unknown optimizer type: {kind}. Must be adam or adamw.")
