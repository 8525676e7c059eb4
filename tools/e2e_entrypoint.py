"""End-to-end timing of the entrypoint's real loop.

scripts/finetune.py feeds `train_step` from `infinite_iter(get_dataloader(...))`: DataLoader workers (token / target
construction, augmentation draws, raw audio) -> pinned, double-buffered H2D on a side stream -> log-mel + SpecAugment on the
device (`GpuMelLoader`) -> `model_utils.train_step`  (reference loop: /root/reference/src/whisper_finetune/scripts/finetune.py
:177-188, loader: data/data_loader.py:469-529).  bench.py times the same `train_step` on clips already resident in HBM; this
script times the loop AS THE ENTRYPOINT RUNS IT, then replays the very same batches from HBM, and reports both:

    python tools/e2e_entrypoint.py [--model large-v3] [--batch 32] [--steps 12] [--workers 16]

`measure()` is also what bench.py calls for its `entrypoint_loop` key (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
sys.path.insert(0, str(ROOT))


def measure(model_name: str = "large-v3", batch: int = 32, steps: int = 12, warmup: int = 3, workers: int | None = None,
            device=None, model=None, opt=None, sched=None) -> dict:
    from bench import build_model
    from whisper_finetune.data.data_loader import SimpleTokenizer, SyntheticDataset, get_dataloader
    from whisper_finetune.model.model_utils import infinite_iter, train_step
    from whisper_finetune.model.optimizer import WftAdamW

    device = device or torch.device("cuda", torch.cuda.current_device())
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 2) // 2))
    if model is None:
        model, dims = build_model(model_name, device)
        model.train()
        opt = WftAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    dims = model.dims
    t_cfg = {"mixed_precision_training": True, "mp_dtype": "bf16", "accum_grad_steps": 1, "max_grad_norm": 1.0,
             "label_smoothing": 0.1, "is_lora_run": False}
    sa = {"apply": True, "time_mask_param": 100, "freq_mask_param": 43, "time_warp_w": 80, "p": 1.0}
    n_batches = warmup + steps
    ds = SyntheticDataset(batch * n_batches, seed=4321, min_seconds=29.9)  # ~30 s clips: the bench's audio-seconds per clip
    loader = get_dataloader(ds, SimpleTokenizer(), batch_size=batch, n_mels=dims.n_mels, device=device, shuffle=True,
                            num_workers=workers, spec_augment=True, spec_augment_params=sa, no_timestamp_training=True,
                            prompt_use_rate=0.0, no_timestamps_rate=1.0, drop_last=True)

    # ---- (0) the loader alone: how fast can the host side hand over batches (no model work)?
    t0 = time.perf_counter()
    n = 0
    for _ in loader:
        n += 1
    torch.cuda.synchronize()
    loader_s_per_batch = (time.perf_counter() - t0) / max(n, 1)

    # ---- (1) the entrypoint's loop
    keep = []

    def tap(it):  # remember the batches (device tensors) for the replay below; the yielded mel is recomputed there
        for mel, y_in, y_out in it:
            keep.append((y_in, y_out))
            yield mel, y_in, y_out

    it = tap(infinite_iter(loader))
    for _ in range(warmup):
        train_step(model, it, opt, sched, t_cfg)
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        train_step(model, it, opt, sched, t_cfg)
        per.append(time.perf_counter() - t1)  # (train_step ends with loss.item(): the step's device work is done)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / steps

    # ---- (2) the same token shapes from HBM: resident clips -> log-mel + SpecAugment -> train_step (what bench.py times)
    from whisper_finetune.data.gpu_frontend import GpuFrontend

    fe = GpuFrontend(dims.n_mels, device, spec_augment=True, spec_augment_params=sa)
    g = torch.Generator(device=device).manual_seed(7)
    audio = torch.randn(batch, 480000, device=device, generator=g) * 0.1
    toks = keep[warmup:warmup + steps]

    def resident():
        i = 0
        while True:
            y_in, y_out = toks[i % len(toks)]
            i += 1
            yield fe(audio, training=True), y_in, y_out

    it2 = resident()
    for _ in range(2):
        train_step(model, it2, opt, sched, t_cfg)
    torch.cuda.synchronize()
    per_r = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        train_step(model, it2, opt, sched, t_cfg)
        per_r.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / steps
    med = lambda v: sorted(v)[len(v) // 2]
    return {
        "workload": f"whisper-{model_name} full fine-tune, {batch} synthetic 30 s clips per step through SyntheticDataset -> DataLoader "
                    f"({workers} workers, pinned) -> GpuMelLoader (prefetch on a side stream) -> train_step",
        "ms_per_step": round(e2e * 1e3, 2), "ms_per_step_median": round(med(per) * 1e3, 2),
        "resident_ms_per_step": round(res * 1e3, 2), "resident_ms_per_step_median": round(med(per_r) * 1e3, 2),
        "ratio_to_resident": round(e2e / res, 4),
        "loader_alone_ms_per_batch": round(loader_s_per_batch * 1e3, 2), "workers": workers, "steps": steps,
        "audio_s_per_s": round(batch * 30.0 / e2e, 1),
        "mean_decoder_len": round(sum(t[0].shape[1] for t in toks) / len(toks), 1),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workers", type=int, default=None)
    a = ap.parse_args()
    print(json.dumps(measure(a.model, a.batch, a.steps, a.warmup, a.workers)), flush=True)
