"""LR schedules against the reference's own get_scheduler output (tests/golden/ref_sched.json, generated in the build
container by importing model/scheduler.py), optimizer factory behaviour."""
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
    with pytest.raises(NotImplementedError):
        get_optimizer(m, {**conf, "muon": True})


@pytest.mark.gpu
def test_wft_adamw_matches_torch_adamw_with_clip_scale():
    torch.manual_seed(0)
    ws = [torch.randn(300, 77), torch.randn(1001)]
    ref = [torch.nn.Parameter(w.clone().cuda()) for w in ws]
    mine = [torch.nn.Parameter(w.clone().cuda()) for w in ws]
    a = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    b = WftAdamW(mine, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    b.grad_scale = torch.tensor([0.5], device="cuda")
    for _ in range(3):
        for r, m in zip(ref, mine):
            g = torch.randn_like(r)
            r.grad = g * 0.5
            m.grad = g.clone()
        a.step(); b.step()
    for r, m in zip(ref, mine):
        torch.testing.assert_close(m, r, atol=1e-6, rtol=1e-5)
