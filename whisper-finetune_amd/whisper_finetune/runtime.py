"""Process/rank runtime for the MI355X build — same public surface as the reference's
`whisper_finetune.runtime` (runtime.py:10-119): module globals RANK / LOCAL_RANK /
WORLD_SIZE / IS_DISTRIBUTED / IS_MAIN, setup_distributed(), barrier(), maybe_no_sync(),
unwrap_model(), print_once(), cleanup() and the rank-0 wandb shims.

One process per GPU (torchrun); backend "nccl" is RCCL on ROCm and runs over xGMI inside a
node.  Differences from the reference, all additive: WFT_DIST_BACKEND=gloo selects the gloo
backend so the multi-process plumbing can be exercised on CPU (tests), and wandb is
imported lazily and only when enabled.
"""
from __future__ import annotations

import contextlib
import threading
import datetime
import os
from typing import Optional

import torch
import torch.distributed as dist

RANK = 0
LOCAL_RANK = 0
WORLD_SIZE = 1
IS_DISTRIBUTED = False
IS_MAIN = True

_wandb = None
_NCCL_TIMEOUT = datetime.timedelta(hours=2)  # rank 0 evaluates while the others wait (runtime.py:29)


def _env_int(name: str, default: int) -> int:
    return int(os.environ.get(name, default))


def setup_distributed() -> torch.device:
    """Initialise the process group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE) and pin
    this process to its GPU.  Returns the device this rank trains on."""
    global RANK, LOCAL_RANK, WORLD_SIZE, IS_DISTRIBUTED, IS_MAIN
    backend = os.environ.get("WFT_DIST_BACKEND", "nccl")
    world = _env_int("WORLD_SIZE", 1)
    IS_DISTRIBUTED = world > 1 and "RANK" in os.environ
    have_gpu = torch.cuda.is_available()
    if backend == "nccl" and not have_gpu:
        raise RuntimeError("This training script requires a ROCm GPU (set WFT_DIST_BACKEND=gloo only for CPU plumbing tests).")
    if IS_DISTRIBUTED:
        RANK, LOCAL_RANK, WORLD_SIZE = _env_int("RANK", 0), _env_int("LOCAL_RANK", 0), world
        if have_gpu:
            torch.cuda.set_device(LOCAL_RANK)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, timeout=_NCCL_TIMEOUT)
    else:
        RANK, LOCAL_RANK, WORLD_SIZE = 0, 0, 1
        if have_gpu:
            torch.cuda.set_device(0)
    IS_MAIN = RANK == 0
    return torch.device("cuda", LOCAL_RANK) if have_gpu else torch.device("cpu")


def is_main() -> bool:
    return IS_MAIN


def print_once(*args, **kwargs) -> None:
    if IS_MAIN:
        print(*args, **kwargs)


def barrier() -> None:
    if not IS_DISTRIBUTED:
        return
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[LOCAL_RANK])
    else:
        dist.barrier()


def cleanup() -> None:
    if IS_DISTRIBUTED and dist.is_initialized():
        dist.destroy_process_group()


def unwrap_model(model: torch.nn.Module) -> torch.nn.Module:
    return getattr(model, "module", model)


def maybe_no_sync(model: torch.nn.Module, enabled: bool):
    """`model.no_sync()` on all but the last micro-batch of an accumulation window
    (model/model_utils.py:63); a null context for un-wrapped models."""
    no_sync = getattr(model, "no_sync", None)
    if enabled and no_sync is not None:
        return no_sync()
    return contextlib.nullcontext()


def ddp_bucket_cap_mb(model, default_mb: int = 64, one_bucket_below_mb: int = 256) -> int:
    """`bucket_cap_mb` for the DDP wrap (reference: scripts/finetune.py:694-705 takes torch's default).  Full fine-tuning moves
    gigabytes of fp32 gradients: 64 MB buckets (xGMI is point-to-point — few, large messages — and the first buckets leave while the
    backward pass still runs).  A LoRA / frozen run moves 56-115 MB in total: latency-bound, ONE bucket — the cap is the
    trainable bytes rounded up."""
    mb = sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad) / 2**20
    return int(mb) + 1 if mb <= one_bucket_below_mb else default_mb


_TLS = threading.local()


def backward_launch_mode() -> int:
    """What the autograd functions of engine/ops.py record at FORWARD time for their backward kernels: 1 (one workgroup per tile /
    item) inside `exchange_launch_mode` on THIS thread, else 0 (persistent grids)."""
    return 1 if getattr(_TLS, "exchange", 0) > 0 else 0


@contextlib.contextmanager
def exchange_launch_mode(active: bool):
    """Launch mode of the backward pass that runs beside a gradient exchange.  `train_step` wraps forward + backward of the LAST
    micro-batch of an accumulation window of a multi-process job in this context (the reference's `no_sync()` window ends there,
    model/model_utils.py:63-72): every autograd function of the engine notes `backward_launch_mode()` in its forward and hands it to
    its backward kernels as a PER-CALL argument (wft_gemm_args.launch_mode / wft_attn_args.launch_mode) — the 256x256 NT GEMMs and
    the dK/dV attention kernel of that backward pass run one workgroup per tile / item, the forward pass, the `no_sync()`
    micro-batches and anything another thread does (an evaluator, a second model) keep their persistent grids.  Nothing is switched
    inside libwft: the mode is thread-local here and travels through the autograd graph.  Measured on one GPU with a side-stream
    kernel that holds CUs the way the collectives do (bench.py `ddp_mode_1gpu`): per-tile launches cost 0.5-0.9 % of a step while
    nothing holds CUs and are 0.6-1.1 % faster than persistent grids while 24 CUs are held."""
    if not active:
        yield
        return
    _TLS.exchange = getattr(_TLS, "exchange", 0) + 1
    try:
        yield
    finally:
        _TLS.exchange -= 1


# ---------------------------------------------------------------- wandb (rank 0 only)
def setup_wandb(**kwargs) -> None:
    global _wandb
    _wandb = None
    if IS_MAIN:
        import wandb  # optional dependency

        wandb.init(**kwargs)
        _wandb = wandb


def log(data: dict, step: Optional[int] = None) -> None:
    if _wandb is not None:
        _wandb.log(data, step=step)


def watch(model: torch.nn.Module, **kwargs) -> None:
    if _wandb is not None:
        _wandb.watch(unwrap_model(model), **kwargs)


def save_wandb_file(path: str) -> None:
    if _wandb is not None:
        _wandb.save(path)


def update_wandb_config(data: dict, **kwargs) -> None:
    if _wandb is not None:
        _wandb.config.update(data, **kwargs)


def set_wandb_summary(key: str, value) -> None:
    if _wandb is not None:
        _wandb.summary[key] = value


def finish_wandb() -> None:
    if _wandb is not None:
        _wandb.finish()
