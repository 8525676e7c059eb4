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


def disable_all_grads(model) -> None:
    """Freeze every parameter of `model` (utils.py:138-140; finetune.py freezes the encoder / decoder half with it)."""
    for p in model.parameters():
        p.requires_grad = False


def print_trainable_parameters(model) -> None:
    """utils.py:129-135."""
    total = trainable = 0
    for p in model.parameters():
        total += p.numel()
        trainable += p.numel() if p.requires_grad else 0
    print(f"Number of trainable parameters: {trainable:,} out of total {total:,}.")


def print_size_of_model(model, label: str = "") -> int:
    """Serialised size of the state dict in bytes (utils.py:120-126); goes through a scratch file like the reference."""
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".p") as fh:
        torch.save(model.state_dict(), fh.name)
        size = os.path.getsize(fh.name)
    print("model: ", label, " \t", "Size (MB):", size / 1e6)
    return size


def handle_cuda_memory_operations(config: dict) -> None:
    """Dump the allocator's memory-history snapshot to memory/memory_<bfloat16>_<lora>_<batch>_<mp>_<dtype>.pt and stop
    recording (utils.py:91-117).  torch.cuda.memory is the HIP caching allocator on ROCm; failures are reported, not raised."""
    m, d, t = config.get("model", {}), config.get("dataset", {}), config.get("training", {})
    tag = "_".join(str(v) for v in ("memory", m.get("bfloat16", "NA"), m.get("lora", "NA"), d.get("batch_size", "NA"),
                                    t.get("mixed_precision_training", "NA"), t.get("mp_dtype", "NA")))
    try:
        os.makedirs("memory", exist_ok=True)
        torch.cuda.memory._dump_snapshot(f"memory/{tag}.pt")
    except Exception as exc:
        print(f"Failed to dump CUDA memory snapshot: {exc}")
    try:
        torch.cuda.memory._record_memory_history(enabled=None)
    except Exception as exc:
        print(f"Failed to stop CUDA memory snapshotting: {exc}")
