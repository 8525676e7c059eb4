"""Data helpers with the reference's names (data/utils.py): load_hf_dataset, process_dataset, TimeWarpAugmenter,
ExtremesFrequencyMasking, pad_or_trim.  The augmenters keep the reference's host RNG draws and hand the drawn
parameters to the `wft_specaug` kernel; dataset IO stays on HF `datasets` (host)."""
from __future__ import annotations

import warnings
from collections import defaultdict
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from whisper_finetune.data.languages import LANGUAGES, TO_LANGUAGE_CODE

N_SAMPLES = 480000
N_FFT = 400
HOP_LENGTH = 160


def load_hf_dataset(path_or_name: str, **kwargs):
    """A local path -> load_from_disk, anything else -> load_dataset (hub)."""
    from datasets import load_dataset, load_from_disk

    p = Path(path_or_name)
    if p.exists():
        print(f"Loading local dataset from: {path_or_name}")
        return load_from_disk(str(p))
    print(f"Loading remote dataset: {path_or_name}")
    return load_dataset(path_or_name, **kwargs)


def _pad_list_with_none(values, target_len: int, label: str) -> list:
    """Per-dataset option lists shorter than the dataset list are extended with None (with a warning) so that zip()
    never silently drops a dataset (data/utils.py:193-203)."""
    out = list(values) if values is not None else []
    if len(out) < target_len:
        missing = target_len - len(out)
        warnings.warn(f"{label} has {len(out)} entries for {target_len} datasets; appending {missing} None value(s) to avoid "
                      "dropping data in zip().", stacklevel=2)
        out += [None] * missing
    return out


def _cast_large_string_columns(dataset):
    """Arrow `large_string` columns -> `string`, so datasets written by different tool versions concatenate
    (data/utils.py:206-222)."""
    from datasets import Features, Value

    feats = dict(dataset.features)
    large = [c for c, f in feats.items() if isinstance(f, Value) and f.dtype == "large_string"]
    if not large:
        return dataset
    print("Casting large_string columns to string for dataset schema alignment.")
    for c in large:
        feats[c] = Value("string")
    return dataset.cast(Features(feats))


def _filter_language_tags(dataset, language_tags, dataset_name):
    """Keep the rows whose `language` is one of `language_tags` (None = keep all) — data/utils.py:225-235."""
    if language_tags is None:
        return dataset
    keep = set(language_tags)
    before = len(dataset)
    print(f"Filtering dataset {dataset_name} to language tag(s): {sorted(keep)}")
    dataset = dataset.filter(lambda batch: [lang in keep for lang in batch["language"]], batched=True)
    print(f"Filtered dataset size: {len(dataset)} (from {before})")
    return dataset


def add_fixed_value(batch, col_name, fixed_value):
    batch[col_name] = [fixed_value] * len(batch["text"])
    return batch


def _normalize_language_value(language) -> str:
    """'German' / ' DE ' -> 'de'; anything that is not a Whisper language code or name raises (data/utils.py:355-368)."""
    if not isinstance(language, str):
        raise ValueError(f"Language value {language!r} is not a string.")
    key = language.strip().lower()
    if key in LANGUAGES:
        return key
    code = TO_LANGUAGE_CODE.get(key)
    if code is None:
        raise ValueError(f"Unsupported language value {language!r}.")
    return code


def normalize_language_values(batch):
    batch["language"] = [_normalize_language_value(v) for v in batch["language"]]
    return batch


def _subsample_indices(dataset, n: int, groupby: Optional[str]):
    """`n` rows per distinct value of `groupby` (with replacement when a group is smaller), else min(n, len) rows without
    replacement — numpy's global RNG, as the reference (data/utils.py:316-332)."""
    if groupby and groupby in dataset.column_names:
        print(f"Performing groupby sampling on column: {groupby}")
        groups = defaultdict(list)
        for i, v in enumerate(dataset[groupby]):
            groups[v].append(i)
        picked = []
        for members in groups.values():
            picked.extend(np.random.choice(members, size=n, replace=len(members) < n))
        return picked
    print("Performing regular random sampling")
    return np.random.choice(len(dataset), size=min(n, len(dataset)), replace=False)


def process_dataset(dataset_names, select_n_per_ds, split_name, groupby_col, print_examples=False, example_count=5,
                    return_sizes=False, select_language_tag=None):
    """Load every listed dataset, pick the split (falling back to 'train', then to the first split), rename `sentence` /
    `sentence_de` to `text`, make sure `language` (normalised; 'de' when absent) and `prompt` ('' when absent) exist, filter
    by language tag BEFORE sub-sampling, sub-sample, align string types, concatenate (data/utils.py:238-352; same positional
    signature).  `print_examples` / `example_count` are accepted and unused, as in the reference."""
    from datasets import concatenate_datasets

    names = list(dataset_names)
    select_n_per_ds = _pad_list_with_none(select_n_per_ds, len(names), "select_n_per_ds")
    groupby_col = _pad_list_with_none(groupby_col, len(names), "groupby_col")
    select_language_tag = ([None] * len(names) if select_language_tag is None
                           else _pad_list_with_none(select_language_tag, len(names), "select_language_tag"))
    parts, sizes = [], []
    for n_sel, groupby, tags, name in zip(select_n_per_ds, groupby_col, select_language_tag, names):
        ds = load_hf_dataset(name)
        if split_name not in ds:
            print(f"Split name {split_name} not found in dataset {name}. Available splits: {list(ds.keys())}")
            split_name = "train" if "train" in ds else list(ds.keys())[0]  # sticky for the following datasets, as upstream
            print(f"Defaulting to split: {split_name}")
        ds = ds[split_name]
        print(f"Processing dataset: {name}")
        print(f"Original dataset size: {len(ds)}")
        for alias in ("sentence", "sentence_de"):
            if alias in ds.column_names:
                ds = ds.rename_column(alias, "text")
        if "language" in ds.column_names:
            ds = ds.map(normalize_language_values, batched=True)
        else:
            ds = ds.map(add_fixed_value, batched=True, fn_kwargs={"col_name": "language", "fixed_value": "de"})
        if "prompt" not in ds.column_names:
            ds = ds.map(add_fixed_value, batched=True, fn_kwargs={"col_name": "prompt", "fixed_value": ""})
        ds = _filter_language_tags(ds, tags, name)
        if n_sel is not None:
            ds = ds.select(_subsample_indices(ds, n_sel, groupby))
            print(f"Number of samples selected: {len(ds)}")
        else:
            print("No sampling performed (N is None)")
        ds = _cast_large_string_columns(ds)
        parts.append(ds)
        sizes.append(len(ds))
    out = concatenate_datasets(parts)
    print(f"Total rows in concatenated dataset: {len(out)}")
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
        if not x.is_cuda:
            raise RuntimeError("ExtremesFrequencyMasking runs on the GPU in this build (wft_specaug); move the spectrogram to the device")
        from whisper_finetune.engine import kernels as K

        ext = torch.zeros((x.shape[0], 2), dtype=torch.int32)
        for b in range(x.shape[0]):
            r = torch.rand(1).item()
            ext[b, 0], ext[b, 1] = int(round(r * self.low_freq_range)), int(round(r * self.high_freq_range))
        out = K.specaug(x.float(), torch.zeros((x.shape[0], 8), dtype=torch.int32, device=x.device), ext.to(x.device))
        return out.squeeze(0) if single else out


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
