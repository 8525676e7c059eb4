#!/usr/bin/env python
"""Fine-tuning entrypoint — same CLI and YAML surface as the reference's scripts/finetune.py:

    torchrun --standalone --nproc_per_node=$N whisper-finetune_amd/whisper_finetune/scripts/finetune.py --config X.yaml

main(config): rt.setup_distributed -> per-rank seed -> global->local accumulation -> model (engine Whisper with
openai-whisper names; optional checkpointed/stochastic encoder/decoder, layer resizing, freezing, LoRA, deep-SpecAugment)
-> datasets -> step arithmetic -> DistributedSampler / WarmupDatasetSampler -> loaders -> optimizer / scheduler ->
DDP (RCCL) -> main_loop: rank-0 evaluation at step 0 and every val_steps with barriers, best_model.pt on a new minimum
macro WER, stepN.pt if save_all_checkpoints, last_model.pt at the end (finetune.py:84-229,310-746).

Additions under the SAME schema (there is no network / dataset / pretrained weight in the build environment):
  model.init_name may also be a path to a {"dims", "model_state_dict"} checkpoint; model.random_init: true builds the
  architecture `init_name` with random weights; dataset.synthetic: {train: N, val: M} swaps the HF datasets for the
  synthetic provider (data/data_loader.SyntheticDataset) and a byte-level tokenizer.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parents[2]
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

import whisper_finetune.runtime as rt  # noqa: E402
from whisper_finetune.data.data_loader import (SimpleTokenizer, SyntheticDataset, WarmupDatasetSampler,  # noqa: E402
                                               get_dataloader, get_dataset_boundary_indices)
from whisper_finetune.engine.whisper_model import MODEL_DIMS, ModelDimensions, Whisper, init_random_  # noqa: E402
from whisper_finetune.eval.evaluator import evaluate_multiple_datasets, log_metrics_to_wandb  # noqa: E402
from whisper_finetune.model.lora import LoRAUpdateTracker, apply_lora, print_lora_info  # noqa: E402
from whisper_finetune.model.model_utils import (CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder,  # noqa: E402
                                                infinite_iter, register_deep_spec_augment_hooks, resize_whisper_layers,
                                                save_model, train_step)
from whisper_finetune.model.optimizer import get_optimizer  # noqa: E402
from whisper_finetune.model.scheduler import get_scheduler  # noqa: E402
from whisper_finetune.utils import (calculate_training_steps, calculate_val_steps, disable_all_grads, get_unique_base_path,  # noqa: E402
                                    read_config, resolve_local_accum_grad_steps, set_seed)

LAYER_PRESETS = {"whisper-4832": ("large-v3", 48, 32), "whisper-3248": ("large-v3", 32, 48)}  # finetune.py:51-54


def resolve_model_spec(m_cfg: dict):
    """init_name (or a preset) -> (base name / checkpoint path, target encoder layers, target decoder layers)."""
    name = m_cfg["init_name"]
    enc = m_cfg.get("encoder_layers", m_cfg.get("encoder_layer"))
    dec = m_cfg.get("decoder_layers", m_cfg.get("decoder_layer", m_cfg.get("deocer_layer")))
    if name in LAYER_PRESETS:
        base, p_enc, p_dec = LAYER_PRESETS[name]
        return m_cfg.get("base_init_name", base), enc or p_enc, dec or p_dec
    return m_cfg.get("base_init_name", name), enc, dec


def build_model(config: dict) -> Whisper:
    m_cfg, t_cfg = config["model"], config["training"]
    base, enc_layers, dec_layers = resolve_model_spec(m_cfg)
    state = None
    if os.path.isfile(base):
        ck = torch.load(base, map_location="cpu")
        dims, state = ModelDimensions(**ck["dims"]), ck["model_state_dict"]
    elif base in MODEL_DIMS:
        dims = ModelDimensions(**vars(MODEL_DIMS[base]))
        if not m_cfg.get("random_init", False):
            raise RuntimeError(f"no pretrained weights for '{base}' are reachable here (no network): point model.init_name at a "
                               "checkpoint file or set model.random_init: true")
    else:
        raise RuntimeError(f"Model {base} not found; available models = {sorted(MODEL_DIMS)}")
    model = Whisper(dims)
    if state is None:
        init_random_(model, seed=config["seed"])
    else:
        model.load_state_dict({k: v.float() for k, v in state.items()})
    # Stochastic depth exists only on the checkpointed encoder / decoder classes, which the reference swaps in when the
    # gradient_checkpointing_* flag of that half is set; the frozen half of an encoder- / decoder-only run gets p = 0
    # (scripts/finetune.py:413-455).
    sd = float(t_cfg.get("stochastic_depth", 0.0))
    enc_p = 0.0 if t_cfg.get("train_only_decoder", False) else sd
    dec_p = 0.0 if t_cfg.get("train_only_encoder", False) else sd
    ckpt_enc = bool(t_cfg.get("gradient_checkpointing_encoder", False))
    ckpt_dec = bool(t_cfg.get("gradient_checkpointing_decoder", False))
    saved = model.state_dict() if (ckpt_enc or ckpt_dec) else None
    if ckpt_enc:
        if t_cfg.get("gradient_checkpointing_encoder_last_only", False):
            raise ValueError("gradient_checkpointing_encoder_last_only is not supported when gradient_checkpointing_encoder is enabled")
        model.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head,
                                                           dims.n_audio_layer, enc_p)
    if ckpt_dec:
        model.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                          dims.n_text_layer, dec_p)
    if saved is not None:
        missing, unexpected = model.load_state_dict(saved, strict=True)
        if missing or unexpected:
            raise RuntimeError(f"Unexpected state-dict mismatch. Missing: {missing}, Unexpected: {unexpected}")
    for part in (model.encoder, model.decoder):
        if hasattr(part, "recompute"):
            # the flag selects the reference's module classes (and with them stochastic depth); whether the kept blocks are
            # RECOMPUTED in the backward is a memory decision: 288 GB of HBM keep the activations resident by default
            part.recompute = bool(t_cfg.get("wft_recompute", False))
    if resize_whisper_layers(model, enc_layers, dec_layers):
        print(f"Whisper architecture override active: encoder={model.dims.n_audio_layer}, decoder={model.dims.n_text_layer}")
    model.is_bfloat = False  # scripts/finetune.py:410: always False, AMP handles precision
    if t_cfg.get("train_only_decoder", False):
        disable_all_grads(model.encoder)
    if t_cfg.get("train_only_encoder", False):
        disable_all_grads(model.decoder)
    config["training"]["is_lora_run"] = bool(m_cfg.get("lora", False))
    if m_cfg.get("lora", False):
        apply_lora(model, m_cfg["lora_config"], train_only_decoder=t_cfg.get("train_only_decoder", False),
                   train_only_encoder=t_cfg.get("train_only_encoder", False))
        if rt.IS_MAIN:
            print_lora_info(model)
    return model


def _evaluate_and_maybe_checkpoint(model, dev_loaders, t_config, save_dir, step, min_wer, save_checkpoints, tokenizer):
    ds_metrics, macro = evaluate_multiple_datasets(rt.unwrap_model(model), dev_loaders, t_config, tokenizer=tokenizer)
    eval_wer = macro["macro_wer"]
    print(f"Initial Macro WER: {eval_wer:.4f}" if step == 0 else f"Step {step}: Macro WER={eval_wer:.4f}")
    log_metrics_to_wandb(ds_metrics, macro, step=step, prefix="val")
    if step > 0 and eval_wer < min_wer:
        min_wer = eval_wer
        save_model(model, f"{save_dir}/best_model.pt")
        print(f"  Saved new best model (WER: {min_wer:.4f})")
    if step > 0 and save_checkpoints:
        save_model(model, f"{save_dir}/step{step}.pt")
    return min(min_wer, eval_wer)


def main_loop(model, train_loader, dev_loaders, optimizer, scheduler, save_dir, t_config, scaler=None, tokenizer=None):
    rt.watch(model, log="all")
    tracker = LoRAUpdateTracker(rt.unwrap_model(model)) if t_config.get("is_lora_run", False) else None
    min_wer = float("inf")
    if rt.IS_MAIN and dev_loaders:
        print("\nRunning initial evaluation...")
        min_wer = _evaluate_and_maybe_checkpoint(model, dev_loaders, t_config, save_dir, 0, min_wer, False, tokenizer)
    rt.barrier()
    train_iter = infinite_iter(train_loader)
    losses = []
    for step in range(1, t_config["train_steps"] + 1):
        loss = train_step(model, train_iter, optimizer, scheduler, t_config, lora_tracker=tracker, step=step, scaler=scaler)
        losses.append(loss)
        rt.print_once(f"step {step}/{t_config['train_steps']} loss {loss:.4f} lr {optimizer.param_groups[0]['lr']:.3e}")
        rt.log({"train/loss": loss, **{f"lr/group_{i}": g["lr"] for i, g in enumerate(optimizer.param_groups)}}, step=step)
        assert loss < t_config["max_train_loss"], f"Train loss is above {t_config['max_train_loss']}, the loss is unable to converge."
        if step % t_config["val_steps"] == 0 or step == t_config["train_steps"]:
            if rt.IS_MAIN and dev_loaders:
                min_wer = _evaluate_and_maybe_checkpoint(model, dev_loaders, t_config, save_dir, step, min_wer,
                                                         t_config.get("save_all_checkpoints", False), tokenizer)
            rt.barrier()
    if rt.IS_MAIN:
        save_model(model, f"{save_dir}/last_model.pt")
    rt.barrier()
    return losses


def resolve_precision(t_cfg: dict) -> str:
    """Map the YAML's precision request onto the engine's two compute modes -> "bf16" | "fp32".
    mixed_precision_training: False is the reference's true-fp32 path (model/model_utils.py:37-48,64) = the engine's fp32
    mode (engine/ops32.py, fp32 MFMA).  Every shipped reference YAML says `mp_dtype: fp16`; on MI355X that becomes bf16
    autocast (bf16 MFMA inputs, fp32 accumulation and master weights) without a GradScaler — said loudly, once, and recorded
    in the config dump."""
    if not t_cfg["mixed_precision_training"]:
        rt.print_once("mixed_precision_training: False -> fp32 compute mode (parity mode: fp32 MFMA kernels, nothing fused)")
        return "fp32"
    if t_cfg["mp_dtype"] == "fp16":
        import warnings

        msg = ("training.mp_dtype: fp16 -> running bf16 autocast on MI355X (fp32 master weights and accumulation, no "
               "GradScaler: bf16 has fp32's exponent range).  Set mp_dtype: bf16 to silence this.")
        warnings.warn(msg, stacklevel=2)
        rt.print_once("WARNING: " + msg)
        t_cfg["wft_requested_mp_dtype"] = "fp16"
        t_cfg["mp_dtype"] = "bf16"
    return "bf16"


def main(config: dict):
    device = rt.setup_distributed()
    set_seed(config["seed"] + rt.RANK)
    t_cfg, d_cfg = config["training"], config["dataset"]
    compute_dtype = resolve_precision(t_cfg)
    if config.get("model", {}).get("bfloat16", False):
        print("WARNING: config['model']['bfloat16'] is deprecated and will be ignored!")
    config["training"]["global_accum_grad_steps"] = t_cfg["accum_grad_steps"]
    t_cfg["accum_grad_steps"] = resolve_local_accum_grad_steps(t_cfg["accum_grad_steps"], rt.WORLD_SIZE)
    config["save_dir"] = os.path.join(config.get("save_dir", "output"), get_unique_base_path())
    if rt.IS_MAIN:
        os.makedirs(config["save_dir"], exist_ok=True)
    rt.barrier()

    model = build_model(config).to(device)
    model.set_compute_dtype(compute_dtype)  # after every module swap (checkpointed classes, resizing, LoRA)
    aug = config.get("augmentation", {})
    dsa = aug.get("deep_spec_augment", {})
    if dsa.get("apply", False):
        register_deep_spec_augment_hooks(model, dsa["time_mask_param"], dsa["freq_mask_param"], p=dsa.get("p", 1.0),
                                         layer_indices=dsa.get("layer_indices"))

    # ---- datasets
    syn = d_cfg.get("synthetic")
    if syn:
        tokenizer = SimpleTokenizer()
        train_ds = SyntheticDataset(int(syn["train"]), seed=config["seed"])
        val_sets = {"synthetic_val": SyntheticDataset(int(syn.get("val", 0)), seed=config["seed"] + 10_000)} if (rt.IS_MAIN and syn.get("val")) else {}
        boundaries = get_dataset_boundary_indices([len(train_ds)])
    else:
        from whisper_finetune.data.utils import _pad_list_with_none, process_dataset  # HF datasets path (disk / network)
        from whisper.tokenizer import get_tokenizer

        tokenizer = get_tokenizer(multilingual=True, language="de", task="transcribe")
        train_ds, sizes = process_dataset(d_cfg["train_datasets"], d_cfg["select_n_per_t_ds"], d_cfg["train_split_name"],
                                          d_cfg["groupby_col"], return_sizes=True, select_language_tag=d_cfg.get("select_language_tag"))
        boundaries = get_dataset_boundary_indices(sizes)
        val_sets = {}
        if rt.IS_MAIN:  # validation sets only exist on rank 0 (scripts/finetune.py:543-576)
            val_paths = d_cfg.get("val_datasets", [])
            val_paths = [val_paths] if isinstance(val_paths, str) else list(val_paths)
            names = d_cfg.get("val_dataset_names")
            if names is None:
                names = [v.split("/")[-1] for v in val_paths]
            else:
                names = _pad_list_with_none(names, len(val_paths), "val_dataset_names")
            sel_n, groupby = d_cfg["select_n_per_v_ds"], d_cfg.get("groupby_col", [])
            for i, (path, name) in enumerate(zip(val_paths, names)):
                n_i = sel_n[i] if i < len(sel_n) else None
                g_i = groupby[i] if i < len(groupby) else None
                val_sets[name] = process_dataset([path], [n_i], d_cfg["valid_split_name"], [g_i])

    # ---- step arithmetic
    drop_last = d_cfg.get("drop_last", True)
    t_cfg["train_steps"] = calculate_training_steps(config, train_ds, rt.WORLD_SIZE, drop_last)
    t_cfg["val_steps"] = calculate_val_steps(config)
    sched_cfg = config["lr_scheduler"]
    if sched_cfg["warmup_steps"] < 1:
        sched_cfg["warmup_steps"] = int(sched_cfg["warmup_steps"] * t_cfg["train_steps"])

    # ---- samplers / loaders
    sampler = None
    warm_idx = d_cfg.get("warmup_dataset_idx")
    if rt.IS_DISTRIBUTED:
        if warm_idx is not None:
            raise ValueError("dataset.warmup_dataset_idx is not supported under DDP")
        sampler = torch.utils.data.distributed.DistributedSampler(train_ds, num_replicas=rt.WORLD_SIZE, rank=rt.RANK, shuffle=True,
                                                                  seed=config["seed"], drop_last=drop_last)
    elif warm_idx is not None:
        lo, hi = boundaries[warm_idx]
        sampler = WarmupDatasetSampler(list(range(lo, hi)), list(range(len(train_ds))),
                                       sched_cfg["warmup_steps"] * t_cfg["accum_grad_steps"], d_cfg["batch_size"])
    sa = aug.get("spec_augment", {})
    ex = aug.get("extremes_spec_augment", {})
    aa = aug.get("audio_augment", {})  # forwarded like the reference does: AudioDataset refuses what is out of scope
    n_mels = rt.unwrap_model(model).dims.n_mels
    loader_kw = dict(n_mels=n_mels, no_timestamp_training=d_cfg.get("no_timestamp_training", False),
                     max_prompt_length=d_cfg.get("max_prompt_length", 223), prompt_use_rate=d_cfg.get("prompt_use_rate", 0.5),
                     no_timestamps_rate=d_cfg.get("no_timestamp_rate", 0.5), device=device)
    train_loader = get_dataloader(train_ds, tokenizer, batch_size=d_cfg["batch_size"], sampler=sampler, shuffle=True,
                                  num_workers=d_cfg.get("train_num_workers", min(os.cpu_count() or 1, 8)),
                                  spec_augment=sa.get("apply", False), spec_augment_params=sa,
                                  extremes_spec_augment=ex.get("apply", False), extremes_spec_augment_params=ex,
                                  apply_baseline_aug=aa.get("apply_baseline_aug", False), apply_office_aug=aa.get("apply_office_aug", False),
                                  apply_advanced_aug=aa.get("apply_advanced_aug", False),
                                  time_stretch_min_rate=aa.get("time_stretch", {}).get("min_rate", 0.8),
                                  time_stretch_max_rate=aa.get("time_stretch", {}).get("max_rate", 1.25),
                                  bpe_dropout=aug.get("bpe_dropout", 0.0), drop_last=drop_last, **loader_kw)
    dev_loaders = {name: get_dataloader(ds, tokenizer, batch_size=d_cfg.get("batch_size_eval", d_cfg["batch_size"]), shuffle=False,
                                        num_workers=d_cfg.get("eval_num_workers", 0), **{**loader_kw, "no_timestamp_training": True,
                                                                                         "prompt_use_rate": 0,
                                                                                         "no_timestamps_rate": 0})
                   for name, ds in val_sets.items()}

    optimizer = get_optimizer(model, config["optimizer"], is_lora_run=t_cfg["is_lora_run"])
    scheduler = get_scheduler(optimizer, sched_cfg, t_cfg["train_steps"])
    use_fp16 = t_cfg["mixed_precision_training"] and t_cfg["mp_dtype"] == "fp16"
    scaler = torch.amp.GradScaler("cuda") if use_fp16 else None

    if rt.IS_DISTRIBUTED:
        from torch.nn.parallel import DistributedDataParallel as DDP

        unused = t_cfg.get("ddp_find_unused_parameters", float(t_cfg.get("stochastic_depth", 0.0)) > 0)
        kw = dict(device_ids=[rt.LOCAL_RANK], output_device=rt.LOCAL_RANK) if device.type == "cuda" else {}
        model = DDP(model, find_unused_parameters=unused, broadcast_buffers=False, gradient_as_bucket_view=True,
                    bucket_cap_mb=rt.ddp_bucket_cap_mb(model), **kw)

    w_cfg = dict(config.get("wandb", {}))
    if w_cfg.pop("enabled", False):
        rt.setup_wandb(**w_cfg, config=config)
    losses = main_loop(model, train_loader, dev_loaders, optimizer, scheduler, config["save_dir"], t_cfg, scaler=scaler,
                       tokenizer=tokenizer)
    rt.finish_wandb()
    if device.type == "cuda":
        rt.print_once(f"Peak memory: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Fine-tune Whisper on MI355X")
    ap.add_argument("--config", type=str, required=True, help="Path to the YAML config")
    args = ap.parse_args()
    cfg = read_config(args.config)
    cfg["path_to_config"] = args.config
    try:
        main(cfg)
    finally:
        rt.cleanup()
