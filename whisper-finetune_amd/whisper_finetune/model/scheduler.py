"""Learning-rate schedules with the reference's `get_scheduler(optimizer, s_conf, train_steps)` surface
(model/scheduler.py:114-151).  Host scalar math only: one multiplier function per schedule type.
linear / cosine / cosine_with_restarts follow the public `transformers` formulas the reference calls;
cosine_with_warmup_restarts(_chill) follow model/scheduler.py:15-71."""
from __future__ import annotations

import math
import random
from functools import partial

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


def _linear(step: int, *, warmup: int, total: int) -> float:
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


def _cosine(step: int, *, warmup: int, total: int, cycles: float = 0.5) -> float:
    if step < warmup:
        return step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * prog)))


def _cosine_hard_restarts(step: int, *, warmup: int, total: int, cycles: int = 1) -> float:
    if step < warmup:
        return step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    if prog >= 1.0:
        return 0.0
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((cycles * prog) % 1.0))))


def _warm_restarts(step: int, *, warmup: int, total: int, cycles: int, gamma: float, chill_steps: int = 0, chill_range: float = 0.0) -> float:
    """Every cycle (total/cycles steps) re-warms linearly to gamma**cycle, then follows the global cosine phase; the
    optional chill phase holds the LR (plus uniform jitter) for the last `chill_steps` of all but the final cycle."""
    prog = (step - warmup) / max(1, total - warmup)
    if prog >= 1.0:
        return 0.0
    cyc_len = total / cycles
    cycle = step // cyc_len
    peak = gamma ** cycle
    in_cycle = step % cyc_len
    if in_cycle < warmup:
        return in_cycle / max(1, warmup) * peak
    if chill_steps and (cyc_len - in_cycle) < chill_steps and cycle < cycles - 1:
        p0 = ((cyc_len - chill_steps + 10) - warmup) / max(1, total - warmup)
        hold = max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((cycles * p0) % 1.0))) * peak)
        return hold + random.uniform(-chill_range, chill_range)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((cycles * prog) % 1.0))) * peak)


def get_cosine_annealing_with_warmup_restarts(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int,
                                              num_cycles: int = 1, last_epoch: int = -1, gamma: float = 1.0) -> LambdaLR:
    """Public constructor with the reference's name and argument order (model/scheduler.py:74-90)."""
    return LambdaLR(optimizer, partial(_warm_restarts, warmup=num_warmup_steps, total=num_training_steps, cycles=num_cycles,
                                       gamma=gamma), last_epoch)


def get_cosine_annealing_with_warmup_restarts_chill(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int,
                                                    num_cycles: int = 1, last_epoch: int = -1, gamma: float = 1.0,
                                                    chill_steps: int = 100, chill_range: float = 0.02) -> LambdaLR:
    """model/scheduler.py:93-111."""
    return LambdaLR(optimizer, partial(_warm_restarts, warmup=num_warmup_steps, total=num_training_steps, cycles=num_cycles,
                                       gamma=gamma, chill_steps=chill_steps, chill_range=chill_range), last_epoch)


def get_scheduler(optimizer: Optimizer, s_conf: dict, train_steps: int) -> LambdaLR:
    kind, warm = s_conf["type"], s_conf["warmup_steps"]
    if kind == "linear":
        fn = partial(_linear, warmup=warm, total=train_steps)
    elif kind == "cosine":
        fn = partial(_cosine, warmup=warm, total=train_steps)
    elif kind == "cosine_with_restarts":
        fn = partial(_cosine_hard_restarts, warmup=warm, total=train_steps, cycles=s_conf["lr_num_cycles"])
    elif kind == "cosine_with_warmup_restarts":
        return get_cosine_annealing_with_warmup_restarts(optimizer, warm, train_steps, num_cycles=s_conf["lr_num_cycles"],
                                                         gamma=s_conf["lr_gamma"])
    elif kind == "cosine_with_warmup_restarts_chill":
        return get_cosine_annealing_with_warmup_restarts_chill(optimizer, warm, train_steps, num_cycles=s_conf["lr_num_cycles"],
                                                               gamma=s_conf["lr_gamma"], chill_steps=s_conf["chill_steps"],
                                                               chill_range=s_conf["chill_range"])
    else:
        raise Exception(f"Unknown learning rate scheduler: {kind}. Must be linear, cosine, cosine_with_restarts or cosine_with_warmup_restarts")
    return LambdaLR(optimizer, fn)
