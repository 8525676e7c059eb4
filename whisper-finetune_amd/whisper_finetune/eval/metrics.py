"""Evaluation metrics with the reference's names (eval/metrics.py): PerUtteranceMetrics / DatasetMetrics,
compute_wer, compute_cer_batch, compute_token_metrics, compute_ece, aggregate_dataset_metrics,
compute_macro_average.  WER / CER are plain Levenshtein rates (jiwer's definition: edits / reference length);
the per-token statistics come from ONE pass of the `wft_token_stats` kernel when logits live on the GPU."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class PerUtteranceMetrics:
    prediction: str
    reference: str
    wer: float
    cer: float
    token_nll: float
    avg_log_prob: float
    token_entropy: float
    token_confidences: List[float] = field(default_factory=list)
    token_correct: List[bool] = field(default_factory=list)


@dataclass
class DatasetMetrics:
    dataset_name: str
    num_samples: int
    wer: float
    cer: float
    mean_token_nll: float
    avg_log_prob: float
    mean_token_entropy: float
    ece: float
    per_utterance: List[PerUtteranceMetrics] = field(default_factory=list)


def _edit_distance(ref: Sequence, hyp: Sequence) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def wer(reference: str, hypothesis: str) -> float:
    """word error rate = (S + D + I) / #reference words."""
    ref = reference.split()
    if not ref:
        raise ValueError("one or more references are empty strings")
    return _edit_distance(ref, hypothesis.split()) / len(ref)


def cer(reference: str, hypothesis: str) -> float:
    """character error rate over the characters of the (stripped) strings, blanks included."""
    ref = list(reference.strip())
    if not ref:
        raise ValueError("one or more references are empty strings")
    return _edit_distance(ref, list(hypothesis.strip())) / len(ref)


def compute_wer(predictions: List[str], references: List[str]) -> List[float]:
    """Per-utterance WER; an empty reference scores 0 for an empty prediction, else 1 (eval/metrics.py:46-62)."""
    out = []
    for pred, ref in zip(predictions, references):
        out.append((0.0 if pred.strip() == "" else 1.0) if ref.strip() == "" else wer(ref, pred))
    return out


def compute_cer_batch(predictions: List[str], references: List[str]) -> List[float]:
    out = []
    for pred, ref in zip(predictions, references):
        out.append((0.0 if pred.strip() == "" else 1.0) if ref.strip() == "" else cer(ref, pred))
    return out


def compute_token_metrics(logits: torch.Tensor, target_ids: torch.Tensor, predicted_ids: torch.Tensor
                          ) -> Tuple[float, float, float, List[float], List[bool]]:
    """(mean NLL, mean log p(predicted token), mean entropy, per-token max-prob, per-token correctness) over the
    rows whose target is not -100; logits [S, V] (eval/metrics.py:85-137)."""
    valid = target_ids != -100
    if valid.sum() == 0:
        return 0.0, 0.0, 0.0, [], []
    lg, tg, pr = logits[valid].float(), target_ids[valid], predicted_ids[valid]
    logp = F.log_softmax(lg, dim=-1)
    p = logp.exp()
    nll = F.cross_entropy(lg, tg, reduction="none").mean().item()
    avg_lp = logp.gather(1, pr.unsqueeze(1)).squeeze(1).mean().item()
    ent = -(p * logp).sum(-1).mean().item()
    return nll, avg_lp, ent, p.max(-1).values.cpu().tolist(), (pr == tg).cpu().tolist()


def token_metrics_from_stats(stats: np.ndarray, argmax: np.ndarray, targets: np.ndarray):
    """Same five quantities from the fused kernel's per-token {lse, max, E_p[x], x_target} (host numpy, one sample).
    The predicted token is the argmax, so log p(pred) = max - lse."""
    valid = targets != -100
    if not valid.any():
        return 0.0, 0.0, 0.0, [], []
    lse, mx, ex, xt = (stats[valid, k].astype(np.float64) for k in range(4))
    return (float(np.mean(lse - xt)), float(np.mean(mx - lse)), float(np.mean(lse - ex)),
            np.exp(mx - lse).tolist(), (argmax[valid] == targets[valid]).tolist())


def compute_ece(all_confidences: List[float], all_correct: List[bool], n_bins: int = 20) -> float:
    """Expected calibration error over equal-width confidence bins (lower, upper]."""
    if len(all_confidences) == 0:
        return 0.0
    conf = np.asarray(all_confidences)
    ok = np.asarray(all_correct, dtype=float)
    edges = np.linspace(0, 1, n_bins + 1)
    ece = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (conf > lo) & (conf <= hi)
        frac = sel.mean()
        if frac > 0:
            ece += frac * abs(conf[sel].mean() - ok[sel].mean())
    return ece


def aggregate_dataset_metrics(per_utterance_metrics: List[PerUtteranceMetrics], dataset_name: str) -> DatasetMetrics:
    """mean over utterances of each scalar; ECE over all tokens of the dataset."""
    ms = per_utterance_metrics
    if not ms:
        return DatasetMetrics(dataset_name, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, [])
    conf = [c for m in ms for c in m.token_confidences]
    ok = [c for m in ms for c in m.token_correct]
    return DatasetMetrics(
        dataset_name=dataset_name, num_samples=len(ms),
        wer=np.mean([m.wer for m in ms]), cer=np.mean([m.cer for m in ms]),
        mean_token_nll=np.mean([m.token_nll for m in ms]), avg_log_prob=np.mean([m.avg_log_prob for m in ms]),
        mean_token_entropy=np.mean([m.token_entropy for m in ms]), ece=compute_ece(conf, ok), per_utterance=ms,
    )


def compute_macro_average(dataset_metrics: List[DatasetMetrics]) -> Dict[str, float]:
    """Unweighted mean over datasets (every dataset counts equally)."""
    keys = {"macro_wer": "wer", "macro_cer": "cer", "macro_mean_token_nll": "mean_token_nll",
            "macro_avg_log_prob": "avg_log_prob", "macro_mean_token_entropy": "mean_token_entropy", "macro_ece": "ece"}
    if not dataset_metrics:
        return {k: 0.0 for k in keys}
    return {k: np.mean([getattr(m, a) for m in dataset_metrics]) for k, a in keys.items()}
