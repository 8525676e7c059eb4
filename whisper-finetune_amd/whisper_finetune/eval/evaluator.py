"""Multi-dataset evaluation (rank 0), same entry points as the reference's eval/evaluator.py:
evaluate_single_dataset (:29-131), evaluate_multiple_datasets (:134-183), log_metrics_to_wandb (:186-221).

The forward is the engine's teacher-forced pass.  Differences that do not change results: the per-token
reductions (argmax, NLL, log-prob, entropy, confidence) come from one fused kernel over the bf16 logits instead
of materialising fp32 [B, S, V] logits plus four softmax passes, and predictions are copied to the host ONCE
per batch instead of once per sample."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

import whisper_finetune.runtime as rt
from whisper_finetune.eval.metrics import (DatasetMetrics, PerUtteranceMetrics, aggregate_dataset_metrics, cer,
                                           compute_macro_average, compute_token_metrics, token_metrics_from_stats, wer)
from whisper_finetune.eval.utils import VOCAB_SPECS, normalize_text


def _default_tokenizer():
    try:
        from whisper.tokenizer import get_tokenizer  # openai-whisper, optional at eval time
    except ImportError as exc:  # pragma: no cover - depends on the environment
        raise RuntimeError("evaluation needs a tokenizer: pass `tokenizer=` or install openai-whisper") from exc
    return get_tokenizer(multilingual=True, language="de", task="transcribe")


def _batch_token_stats(model, x, y_in, y_out):
    """-> (argmax i64 [B,S], stats f32 [B,S,4]) on the host; fused kernel when the model is the engine's."""
    core = rt.unwrap_model(model)
    if hasattr(core, "decoder") and hasattr(core.decoder, "padded_logits") and x.is_cuda and getattr(core, "compute_dtype", "bf16") == "bf16":
        from whisper_finetune.engine import kernels as K

        h = core.decoder.hidden(y_in, core.encoder(x))
        padded = core.decoder.padded_logits(h)
        stats, am = K.token_stats(padded, y_out.reshape(-1), core.dims.n_vocab)
        B, S = y_in.shape
        return am.view(B, S).cpu().numpy(), stats.view(B, S, 4).cpu().numpy()
    return None


@torch.no_grad()
def evaluate_single_dataset(model, dataloader, dataset_name: str, t_config: dict, tokenizer=None) -> DatasetMetrics:
    model = rt.unwrap_model(model)
    model.eval()
    device = next(model.parameters()).device
    mixed = t_config.get("mixed_precision_training", True)
    from whisper_finetune.engine.whisper_model import Whisper as _EngineWhisper, check_amp_request

    mp_dtype = t_config.get("mp_dtype", "fp16")
    if mixed and mp_dtype == "fp16" and isinstance(model, _EngineWhisper):
        # the reference's evaluator accepts a minimal config (mp_dtype defaults to fp16, every shipped YAML says fp16): on the
        # engine that means bf16 autocast, said once — the same rewrite scripts/finetune.resolve_precision makes for training
        rt.print_once("WARNING: evaluation with mp_dtype: fp16 -> bf16 autocast on MI355X (the libwft engine computes in bf16; "
                      "set mp_dtype: bf16 to silence this)")
        mp_dtype = "bf16"
    amp_dtype = torch.float16 if mp_dtype == "fp16" else torch.bfloat16
    check_amp_request(model, mixed, mp_dtype)
    if tokenizer is None:
        tokenizer = _default_tokenizer()
    specials = set(tokenizer.special_tokens.values())
    spec = VOCAB_SPECS["v0"]
    per_utt: List[PerUtteranceMetrics] = []

    for x, y_in, y_out in dataloader:
        x = x.to(device, non_blocking=True)
        y_in = y_in.to(device, non_blocking=True)
        y_out = y_out.to(device, non_blocking=True)
        with torch.autocast(device_type=device.type, enabled=mixed, dtype=amp_dtype):
            fused = _batch_token_stats(model, x, y_in, y_out)
            if fused is None:
                logits = model(x, y_in)
                pred = torch.argmax(logits, dim=-1)
        y_host = y_out.cpu().numpy()
        pred_host = fused[0] if fused is not None else pred.cpu().numpy()
        for i in range(y_host.shape[0]):
            pred_tokens = [t for t in pred_host[i].tolist() if t not in specials and t != -100]
            true_tokens = [t for t in y_host[i].tolist() if t not in specials and t != -100]
            true_text = tokenizer.decode(true_tokens)
            if true_text.strip() == "":
                continue  # empty references are skipped
            pred_n = normalize_text(tokenizer.decode(pred_tokens), **spec)
            true_n = normalize_text(true_text, **spec)
            if fused is not None:
                nll, lp, ent, conf, ok = token_metrics_from_stats(fused[1][i], pred_host[i], y_host[i])
            else:
                nll, lp, ent, conf, ok = compute_token_metrics(logits[i], y_out[i], pred[i])
            per_utt.append(PerUtteranceMetrics(pred_n, true_n, wer(true_n, pred_n) if true_n else 0.0,
                                               cer(true_n, pred_n) if true_n else 0.0, nll, lp, ent, conf, ok))
    return aggregate_dataset_metrics(per_utt, dataset_name)


@torch.no_grad()
def evaluate_multiple_datasets(model, dataloaders: Dict[str, object], t_config: dict, tokenizer=None
                               ) -> Tuple[Dict[str, DatasetMetrics], Dict[str, float]]:
    """-> ({name: DatasetMetrics}, macro averages).  `macro_wer` drives best-checkpoint selection."""
    if tokenizer is None:
        tokenizer = _default_tokenizer()
    results = {}
    for name, loader in dataloaders.items():
        results[name] = evaluate_single_dataset(model, loader, name, t_config, tokenizer)
        m = results[name]
        rt.print_once(f"  {name}: n={m.num_samples} WER={m.wer:.4f} CER={m.cer:.4f} NLL={m.mean_token_nll:.4f} ECE={m.ece:.4f}")
    macro = compute_macro_average(list(results.values()))
    return results, macro


def log_metrics_to_wandb(dataset_metrics: Dict[str, DatasetMetrics], macro_metrics: Dict[str, float], step: int,
                         prefix: str = "val") -> None:
    data = {}
    for name, m in dataset_metrics.items():
        for key in ("wer", "cer", "mean_token_nll", "avg_log_prob", "mean_token_entropy", "ece", "num_samples"):
            data[f"{prefix}/{name}/{key}"] = getattr(m, key)
    for key, val in macro_metrics.items():
        data[f"{prefix}/{key}"] = val
    rt.log(data, step=step)
