"""Data helpers with the reference's names (data/utils.py): load_hf_dataset, process_dataset, TimeWarpAugmenter,
ExtremesFrequencyMasking, pad_or_trim.  The augmenters keep the reference's host RNG draws and hand the drawn
parameters to the `wft_specaug` kernel; dataset IO stays on HF `datasets` (host)."""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

N_SAMPLES = 480000


def load_hf_dataset(path_or_name: str, **kwargs):
    """A local path -> load_from_disk, anything else -> load_dataset (hub)."""
    from datasets import load_dataset, load_from_disk

    p = Path(path_or_name)
    if p.exists():
        print(f"Loading local dataset from: {path_or_name}")
        return load_from_disk(str(p))
    print(f"Loading remote dataset: {path_or_name}")
    return load_dataset(path_or_name, **kwargs)


def process_dataset(dataset_names: Sequence[str], select_n_per_ds: Optional[Sequence[Optional[int]]] = None, split_name: str = "train",
                    groupby_col: Optional[Sequence[Optional[str]]] = None, select_language_tag: Optional[Sequence[Optional[str]]] = None,
                    return_sizes: bool = False):
    """Load, optionally filter by language / subsample, cast to the common schema {audio, text, language, prompt} and
    concatenate the listed datasets (data/utils.py:238-352)."""
    from datasets import Value, concatenate_datasets

    n = len(dataset_names)
    pad = lambda xs: list(xs or []) + [None] * (n - len(xs or []))  # noqa: E731
    parts, sizes = [], []
    for name, n_sel, _grp, lang in zip(dataset_names, pad(select_n_per_ds), pad(groupby_col), pad(select_language_tag)):
        ds = load_hf_dataset(name)
        if hasattr(ds, "keys") and split_name in ds:
            ds = ds[split_name]
        if lang is not None and "language" in ds.column_names:
            ds = ds.filter(lambda r: r["language"] == lang)
        if "language" not in ds.column_names:
            ds = ds.add_column("language", ["de"] * len(ds))
        if "prompt" not in ds.column_names:
            ds = ds.add_column("prompt", [""] * len(ds))
        if n_sel is not None and n_sel < len(ds):
            ds = ds.shuffle(seed=42).select(range(n_sel))
        ds = ds.cast_column("text", Value("string")).select_columns(["audio", "text", "language", "prompt"])
        parts.append(ds)
        sizes.append(len(ds))
    out = concatenate_datasets(parts) if len(parts) > 1 else parts[0]
    return (out, sizes) if return_sizes else out


class TimeWarpAugmenter:
    """SpecAugment time warp: warp_p ~ randint(W, L-W), warp_d ~ randint(-W, W), cubic Hermite through
    (0,-1), (warp_p, .), (L-1, 1), bilinear resampling along time (data/utils.py:41-143) — evaluated by wft_specaug."""

    def __init__(self, W: int = 50):
        self.W = W

    def __call__(self, specs):
        if not torch.is_tensor(specs):
            specs = torch.from_numpy(specs)
        if specs.dim() < 2 or specs.dim() > 3:
            raise ValueError("You sure it's a Spectrogram?")
        single = specs.dim() == 2
        x = specs.unsqueeze(0) if single else specs
        if not x.is_cuda:
            raise RuntimeError("TimeWarpAugmenter runs on the GPU in this build (wft_specaug); move the spectrogram to the device")
        from whisper_finetune.engine import kernels as K

        B, _, L = x.shape
        warp_p = torch.randint(self.W, L - self.W, (B,))
        warp_d = torch.randint(-self.W, self.W, (B,))
        params = torch.zeros((B, 8), dtype=torch.int32)
        params[:, 0], params[:, 1], params[:, 2] = 1, warp_p.int(), warp_d.int()
        out = K.specaug(x.float(), params.to(x.device))
        return out.squeeze(0) if single else out


class ExtremesFrequencyMasking:
    """Zero the lowest round(r*low) and highest round(r*high) mel bins, one r ~ U(0,1) per sample (data/utils.py:146-190)."""

    def __init__(self, low_freq_range: int = 10, high_freq_range: int = 10):
        self.low_freq_range, self.high_freq_range = low_freq_range, high_freq_range

    def __call__(self, specs: torch.Tensor) -> torch.Tensor:
        if not torch.is_tensor(specs):
            specs = torch.tensor(specs)
        single = specs.dim() == 2
        x = specs.unsqueeze(0) if single else specs
        n_mels = x.shape[1]
        for b in range(x.shape[0]):
            r = torch.rand(1).item()
            lo, hi = int(round(r * self.low_freq_range)), int(round(r * self.high_freq_range))
            if lo > 0:
                x[b, : min(lo, n_mels)] = 0
            if hi > 0:
                x[b, max(n_mels - hi, 0):] = 0
        return x.squeeze(0) if single else x


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Trim to `length` or pad with the array's MINIMUM value (silence in the log-mel domain; data/utils.py:380-404)."""
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            widths = [(0, 0)] * array.ndim
            widths[axis] = (0, length - array.shape[axis])
            array = F.pad(array, [p for w in widths[::-1] for p in w], value=torch.min(array).item())
        return array
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - array.shape[axis])
        array = np.pad(array, widths, constant_values=np.min(array))
    return array
