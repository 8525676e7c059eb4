"""Host-side step arithmetic and small helpers with the reference's names (utils.py:14-88)."""
from __future__ import annotations

import math
import os
import random
from datetime import datetime
from typing import Dict

import numpy as np
import torch
import yaml


def calculate_training_steps(config: Dict, train_dataset, world_size: int = 1, drop_last: bool = True) -> int:
    """Optimizer steps for the run (utils.py:14-31): with drop_last
    floor((len // world // batch) * epochs / accum) (at least 1), else
    ceil(len * epochs / (batch * world * accum)).  `accum` is the LOCAL window."""
    n = len(train_dataset)
    world = max(int(world_size), 1)
    epochs = config["training"]["epochs"]
    batch = config["dataset"]["batch_size"]
    accum = config["training"]["accum_grad_steps"]
    if not drop_last:
        return math.ceil(n * epochs / (batch * world * accum))
    micro_per_epoch = (n // world) // batch
    return max(math.floor(micro_per_epoch * epochs / accum), 1)


def resolve_local_accum_grad_steps(accum_grad_steps: int, world_size: int = 1) -> int:
    """The YAML's accum_grad_steps is the GLOBAL window; each rank accumulates global/world
    micro-batches (utils.py:34-48)."""
    accum, world = int(accum_grad_steps), max(int(world_size), 1)
    if accum < 1:
        raise ValueError(f"accum_grad_steps must be >= 1, got {accum}.")
    if accum % world:
        raise ValueError(
            "training.accum_grad_steps is interpreted as the global accumulation window and must be "
            f"divisible by WORLD_SIZE. Got accum_grad_steps={accum} and WORLD_SIZE={world}."
        )
    return accum // world


def calculate_val_steps(config: Dict) -> int:
    t = config["training"]
    return max(int(t["train_steps"] / t["epochs"] * t["eval_steps"]), 1)


def read_config(yaml_file_path):
    print(f"Reading config {yaml_file_path}")
    with open(yaml_file_path, "r") as fh:
        return yaml.safe_load(fh)


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def get_unique_base_path() -> str:
    job = os.environ.get("SLURM_JOB_ID")
    return job if job else datetime.now().strftime("%Y%m%d_%H%M%S")
