"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Restatement of minLoRA (cccntu/minLoRA, git main; un-vendored dependency of the
reference: pyproject.toml:23, imported at model/lora.py:45 and by tests/test_lora.py).

minLoRA is not installed in the build image and is not part of /root/reference, so its published algorithm is restated
here: a `torch.nn.utils.parametrize` parametrization of `weight`,

    W_eff = W + (lora_B @ (lora_A * dropout(ones[1, fan_in]))).view(W.shape) * (alpha / rank)

The adapter arithmetic itself is `oracle.whisper_oracle.lora_effective_weight` — the function the engine's LoRA kernels are
compared with — so the reference's own `tests/test_lora.py` (run by tools/run_reference_tests.sh with this directory on
PYTHONPATH) pins that restatement: A/B shapes, B = 0 => unchanged output, enable/disable, state-dict key names,
merge == parametrized forward, scaling == alpha / rank.

Only tools/run_reference_tests.sh and tests/ put this directory on sys.path; the product package never imports it.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Callable, Dict, Iterator, Optional

import torch
import torch.nn.utils.parametrize as parametrize
from torch import nn

try:
    from oracle.whisper_oracle import lora_effective_weight
except ImportError:  # directory put on sys.path on its own: resolve the repo root from this file
    import os
    import sys

    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
    from oracle.whisper_oracle import lora_effective_weight


class LoRAParametrization(nn.Module):
    def __init__(self, fan_in: int, fan_out: int, fan_in_fan_out: bool = False, rank: int = 4, lora_dropout_p: float = 0.0,
                 lora_alpha: float = 1):
        super().__init__()
        self.fan_in_fan_out = fan_in_fan_out  # embeddings store [in, out]: A and B swap roles
        a_shape, b_shape, m_shape = (rank, fan_in), (fan_out, rank), (1, fan_in)
        if fan_in_fan_out:
            a_shape, b_shape, m_shape = a_shape[::-1], b_shape[::-1], m_shape[::-1]
        self.lora_A = nn.Parameter(torch.zeros(a_shape))
        self.lora_B = nn.Parameter(torch.zeros(b_shape))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        self.lora_alpha, self.rank = lora_alpha, rank
        self.scaling = lora_alpha / rank
        self.lora_dropout_p = lora_dropout_p
        self.lora_dropout = nn.Dropout(p=lora_dropout_p) if lora_dropout_p > 0 else nn.Identity()
        self.register_buffer("lora_dropout_mask", torch.ones(m_shape, dtype=self.lora_A.dtype))
        self._enabled = True

    def forward(self, W: torch.Tensor) -> torch.Tensor:
        if not self._enabled:
            return W
        mask = self.lora_dropout(self.lora_dropout_mask) if self.lora_dropout_p > 0 else None
        if self.fan_in_fan_out:
            A = self.lora_A if mask is None else self.lora_A * mask
            return W + (A @ self.lora_B).view(W.shape) * self.scaling
        return lora_effective_weight(W, self.lora_A, self.lora_B, self.scaling, mask)

    def disable_lora(self) -> None:
        self._enabled = False

    def enable_lora(self) -> None:
        self._enabled = True

    @classmethod
    def from_linear(cls, layer: nn.Module, rank: int = 4, lora_dropout_p: float = 0.0, lora_alpha: float = 1):
        fan_out, fan_in = layer.weight.shape
        return cls(fan_in, fan_out, fan_in_fan_out=False, rank=rank, lora_dropout_p=lora_dropout_p, lora_alpha=lora_alpha)

    @classmethod
    def from_conv2d(cls, layer: nn.Module, rank: int = 4, lora_dropout_p: float = 0.0, lora_alpha: float = 1):
        fan_out, fan_in = layer.weight.view(layer.weight.shape[0], -1).shape
        return cls(fan_in, fan_out, fan_in_fan_out=False, rank=rank, lora_dropout_p=lora_dropout_p, lora_alpha=lora_alpha)

    @classmethod
    def from_embedding(cls, layer: nn.Module, rank: int = 4, lora_dropout_p: float = 0.0, lora_alpha: float = 1):
        fan_in, fan_out = layer.weight.shape
        return cls(fan_in, fan_out, fan_in_fan_out=True, rank=rank, lora_dropout_p=lora_dropout_p, lora_alpha=lora_alpha)


default_lora_config = {nn.Linear: {"weight": partial(LoRAParametrization.from_linear, rank=4)}}


def _adapters(layer: nn.Module) -> Iterator[LoRAParametrization]:
    plists = getattr(layer, "parametrizations", None)
    if plists is None:
        return
    for plist in plists.values():
        for item in plist:
            if isinstance(item, LoRAParametrization):
                yield item


def add_lora(model: nn.Module, lora_config: Optional[Dict] = None) -> None:
    """Register an adapter on every attribute named in lora_config[type(layer)] of every sub-module (exact type match)."""
    cfg = default_lora_config if lora_config is None else lora_config
    for layer in list(model.modules()):
        for attr, make in cfg.get(type(layer), {}).items():
            parametrize.register_parametrization(layer, attr, make(layer))


def apply_to_lora(fn: Callable[[LoRAParametrization], None]) -> Callable[[nn.Module], None]:
    def visit(layer: nn.Module) -> None:
        if isinstance(layer, LoRAParametrization):
            fn(layer)
    return visit


def enable_lora(model: nn.Module) -> None:
    model.apply(apply_to_lora(lambda a: a.enable_lora()))


def disable_lora(model: nn.Module) -> None:
    model.apply(apply_to_lora(lambda a: a.disable_lora()))


def name_is_lora(name: str) -> bool:
    parts = name.split(".")
    return len(parts) >= 4 and parts[-4] == "parametrizations" and parts[-1] in ("lora_A", "lora_B")


def name_is_bias(name: str) -> bool:
    return name.split(".")[-1] == "bias"


def get_params_by_name(model: nn.Module, print_shapes: bool = False, name_filter: Optional[Callable[[str], bool]] = None):
    for name, p in model.named_parameters():
        if name_filter is None or name_filter(name):
            if print_shapes:
                print(name, p.shape)
            yield p


def get_lora_params(model: nn.Module, print_shapes: bool = False):
    return get_params_by_name(model, print_shapes=print_shapes, name_filter=name_is_lora)


def get_bias_params(model: nn.Module, print_shapes: bool = False):
    return get_params_by_name(model, print_shapes=print_shapes, name_filter=name_is_bias)


def get_lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in model.state_dict().items() if name_is_lora(k)}


def _unparametrize(model: nn.Module, keep_effective: bool) -> None:
    for layer in list(model.modules()):
        if any(True for _ in _adapters(layer)):
            for attr in list(layer.parametrizations.keys()):
                parametrize.remove_parametrizations(layer, attr, leave_parametrized=keep_effective)


def merge_lora(model: nn.Module) -> None:
    """Fold every adapter into its weight (W <- W_eff) and drop the parametrization."""
    _unparametrize(model, keep_effective=True)


def remove_lora(model: nn.Module) -> None:
    """Drop the adapters, restoring the original weights."""
    _unparametrize(model, keep_effective=False)
