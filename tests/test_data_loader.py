"""Decoder target construction, prompts, timestamps, collate, samplers (reference: tests/test_data_loader.py:85-171)."""
import numpy as np
import pytest
import torch

from whisper_finetune.data.data_loader import (AudioDataset, SimpleTokenizer, SyntheticDataset, WarmupDatasetSampler, collate_fn,
                                               get_dataloader, get_dataset_boundary_indices)


class _DS(list):
    column_names = ["audio", "text", "language", "prompt"]


def _rec(text, prompt="", n=16000):
    return {"audio": {"array": np.zeros(n, dtype=np.float32)}, "text": text, "language": "de", "prompt": prompt}


def _ds(records, **kw):
    return AudioDataset(_DS(records), SimpleTokenizer(), **kw)


def test_targets_plain_text_no_timestamps():
    tok = SimpleTokenizer()
    ds = _ds([_rec("ab")], no_timestamp_training=True, prompt_use_rate=0.0)
    audio, y_in, y_out, params, ext, cut = ds[0]
    assert audio.shape == (480000,) and cut == 3000
    assert y_in.tolist() == [tok.sot, 50261, 50359, tok.no_timestamps, 97, 98]
    assert y_out.tolist() == [50261, 50359, tok.no_timestamps, 97, 98, tok.eot]
    assert params.tolist() == [0] * 8 and ext.tolist() == [0, 0]


def test_targets_empty_text_gets_nospeech():
    tok = SimpleTokenizer()
    _, y_in, y_out, *_ = _ds([_rec("")], no_timestamp_training=True, prompt_use_rate=0.0)[0]
    assert y_in.tolist() == [tok.sot, 50261, 50359, tok.no_timestamps, tok.no_speech]
    assert y_out.tolist()[-1] == tok.eot


def test_prompt_is_masked_with_minus_100():
    tok = SimpleTokenizer()
    ds = _ds([_rec("ab", prompt="xyz")], no_timestamp_training=True, prompt_use_rate=1.0)
    _, y_in, y_out, *_ = ds[0]
    assert y_in.tolist() == [tok.sot_prev, 120, 121, 122, tok.sot, 50261, 50359, tok.no_timestamps, 97, 98]
    assert y_out.tolist() == [-100, -100, -100, tok.sot, 50261, 50359, tok.no_timestamps, 97, 98, tok.eot]


def test_timestamps_tokens_and_partial_segment_cut():
    tok = SimpleTokenizer()
    ds = _ds([_rec("<|0.00|>a<|1.50|><|2.00|>")], no_timestamp_training=False, no_timestamps_rate=0.0, prompt_use_rate=0.0)
    _, y_in, _, _, _, cut = ds[0]
    tb = tok.timestamp_begin
    assert y_in.tolist() == [tok.sot, 50261, 50359, tb, 97, tb + 75, tb + 100] and cut == 3000  # with timestamps: no cut
    ds2 = _ds([_rec("<|0.00|>a<|1.50|><|2.00|>")], no_timestamp_training=True, prompt_use_rate=0.0)
    _, y_in2, _, _, _, cut2 = ds2[0]
    assert y_in2.tolist() == [tok.sot, 50261, 50359, tok.no_timestamps, 97] and cut2 == 200  # 2.00 s * 100 frames/s
    with pytest.raises(ValueError):
        _ds([_rec("<|0.01|>a")], no_timestamp_training=True)[0]


def test_invalid_records_are_skipped_lazily():
    bad = {"audio": {"array": None}, "text": 5, "language": "de", "prompt": ""}
    ds = _ds([bad, _rec("ok")], no_timestamp_training=True, prompt_use_rate=0.0)
    _, y_in, *_ = ds[0]
    assert 0 in ds.invalid_indices and y_in.tolist()[-2:] == [111, 107]
    with pytest.raises(RuntimeError):
        _ds([bad], no_timestamp_training=True)[0]


def test_collate_and_boundaries_and_warmup_sampler():
    x, yi, yo = collate_fn([(torch.ones(2, 5), torch.tensor([1, 2, 3]), torch.tensor([2, 3, 9])),
                            (torch.ones(2, 5), torch.tensor([4]), torch.tensor([9]))])
    assert yi.tolist() == [[1, 2, 3], [4, 0, 0]] and yo.tolist() == [[2, 3, 9], [9, -100, -100]] and x.shape == (2, 2, 5)
    assert get_dataset_boundary_indices([1000, 500, 2000]) == [(0, 1000), (1000, 1500), (1500, 3500)]
    s = WarmupDatasetSampler([0, 1], list(range(10)), warmup_steps=3, batch_size=2, shuffle=True)
    it = iter(s)
    first = [next(it) for _ in range(6)]
    assert set(first) <= {0, 1}
    later = [next(it) for _ in range(40)]
    assert set(later) - {0, 1}
    with pytest.raises(ValueError):
        WarmupDatasetSampler([], [1], warmup_steps=1, batch_size=1)


def test_cpu_loader_returns_raw_audio_batches_with_drawn_params():
    torch.manual_seed(3)
    loader = get_dataloader(SyntheticDataset(4), SimpleTokenizer(), batch_size=2, n_mels=80, shuffle=False, device=torch.device("cpu"),
                            no_timestamp_training=True, prompt_use_rate=0.0, spec_augment=True,
                            spec_augment_params={"time_mask_param": 100, "freq_mask_param": 27, "time_warp_w": 80, "p": 1.0})
    audio, y_in, y_out, params, ext, cut = next(iter(loader))
    assert audio.shape == (2, 480000) and params.shape == (2, 8) and (params[:, 0] == 1).all()
    assert ((params[:, 1] >= 80) & (params[:, 1] < 2920)).all() and ((params[:, 2] >= -80) & (params[:, 2] < 80)).all()
    assert (y_out[:, -1] == 50257).any() or (y_out == -100).any()
