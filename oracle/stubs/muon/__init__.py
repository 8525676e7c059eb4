"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Restatement of the `muon` package (github.com/KellerJordan/Muon, git HEAD;
un-vendored dependency of the reference: pyproject.toml:29, imported at model/optimizer.py:171).

Optimizer classes with the package's names and param-group contract, stepping through
`oracle.whisper_oracle.muon_with_aux_adam_step` (the restatement the engine's WftMuonWithAuxAdam is compared with).
`MuonWithAuxAdam` is the distributed variant: Muon parameters are sorted by size (largest first), padded to a multiple of
the world size, parameter i of every round is updated by rank i % world and the updated parameter is all-gathered
(SURVEY.md §2.2 C6).  PARITY UNPINNED: the package is neither in /root/reference nor installed, and the reference's
tests only pin the param-group key sets (tests/test_optimizer.py).

Only tools/run_reference_tests.sh and tests/ put this directory on sys.path; the product package never imports it.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

try:
    from oracle import whisper_oracle as O
except ImportError:
    import os
    import sys

    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
    from oracle import whisper_oracle as O

zeropower_via_newtonschulz5 = O.zeropower_via_newtonschulz5
muon_update = O.muon_update
adam_update = O.adam_update

_MUON_KEYS = {"params", "lr", "momentum", "weight_decay", "use_muon"}
_ADAM_KEYS = {"params", "lr", "betas", "eps", "weight_decay", "use_muon"}


def _normalise_groups(param_groups, sort_muon: bool):
    for group in param_groups:
        assert "use_muon" in group
        if group["use_muon"]:
            if sort_muon:
                group["params"] = sorted(group["params"], key=lambda p: p.numel(), reverse=True)
            group.setdefault("lr", 0.02)
            group.setdefault("momentum", 0.95)
            group.setdefault("weight_decay", 0)
            assert set(group.keys()) == _MUON_KEYS
        else:
            group.setdefault("lr", 3e-4)
            group.setdefault("betas", (0.9, 0.95))
            group.setdefault("eps", 1e-10)
            group.setdefault("weight_decay", 0)
            assert set(group.keys()) == _ADAM_KEYS
    return param_groups


class SingleDeviceMuonWithAuxAdam(torch.optim.Optimizer):
    def __init__(self, param_groups):
        super().__init__(_normalise_groups(param_groups, sort_muon=False), dict())
        self._oracle_state = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [{**g, "params": [(p, p.grad) for p in g["params"]]} for g in self.param_groups]
        O.muon_with_aux_adam_step(groups, self._oracle_state)
        return loss


class MuonWithAuxAdam(torch.optim.Optimizer):
    def __init__(self, param_groups):
        super().__init__(_normalise_groups(param_groups, sort_muon=True), dict())
        self._oracle_state = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        world, rank = dist.get_world_size(), dist.get_rank()
        for group in self.param_groups:
            if not group["use_muon"]:
                O.muon_with_aux_adam_step([{**group, "params": [(p, p.grad) for p in group["params"]]}], self._oracle_state)
                continue
            params = group["params"]
            padded = params + [torch.empty_like(params[-1])] * (-len(params) % world)
            for base in range(0, len(padded), world):
                if base + rank < len(params):
                    p = params[base + rank]
                    O.muon_with_aux_adam_step([{**group, "params": [(p, p.grad)]}], self._oracle_state)
                dist.all_gather(padded[base:base + world], padded[base + rank])
        return loss
