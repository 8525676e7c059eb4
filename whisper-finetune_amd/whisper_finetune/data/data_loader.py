"""Data path with the reference's names (data/data_loader.py): AudioDataset, collate_fn, WarmupDatasetSampler,
get_dataset_boundary_indices, get_dataloader — re-cut for the GPU front end.

What moved: the reference computes log-mel + SpecAugment per clip on the CPU inside DataLoader workers and ships
[n_mels, 3000] fp32 mels; here a worker only builds the token sequences, zero-pads the raw audio to 30 s and DRAWS the
augmentation parameters (same default-generator draw order as `AudioDataset.__getitem__` / `_calculate_mel`,
data_loader.py:322-359,273-301), and the main process turns a pinned batch of raw clips into mels with two kernels
(`wft_logmel`, `wft_specaug`).  `get_dataloader(...)` still yields `(mel, y_in, y_out)` batches: mel f32
[B, n_mels, 3000] (on the GPU), y_in padded with 0, y_out padded with -100.
"""
from __future__ import annotations

import re
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset

from whisper_finetune.data import transforms as T
from whisper_finetune.data.gpu_frontend import HOP_LENGTH, N_FFT, N_FRAMES, N_SAMPLES, GpuFrontend, draw_clip_params, mel_filters
from whisper_finetune.data.utils import ExtremesFrequencyMasking, TimeWarpAugmenter, pad_or_trim

CHUNK_LENGTH = 30
_TS = re.compile(r"(<\|[123]?[0-9]\.[0-9][0-9]\|>)")  # <|0.00|> .. <|30.00|>
_FILTERS = {}


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0, device=None) -> torch.Tensor:
    """`whisper.audio.log_mel_spectrogram` (called at data/data_loader.py:278) on the `wft_logmel`
    kernel: audio f32 [n] (numpy or tensor, n a multiple of 160) -> f32 [n_mels, n/160] on the device.  There is no
    host implementation: without a GPU this raises (the batched product path is GpuMelLoader below)."""
    from whisper_finetune.engine import kernels as K
    from whisper_finetune.engine.lib import WftError

    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.asarray(audio, dtype=np.float32))
    if device is None and not audio.is_cuda:
        if not torch.cuda.is_available():
            raise WftError("log_mel_spectrogram runs on the GPU in this build (wft_logmel) and no device is visible")
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None:
        audio = audio.to(device)
    if padding > 0:
        audio = torch.nn.functional.pad(audio, (0, padding))
    if audio.dim() != 1 or audio.shape[0] % HOP_LENGTH:
        raise ValueError(f"expected a 1-D clip whose length is a multiple of {HOP_LENGTH}, got {tuple(audio.shape)}")
    key = (n_mels, audio.device)
    if key not in _FILTERS:
        _FILTERS[key] = mel_filters(n_mels).to(audio.device)
    return K.logmel(audio.float()[None], _FILTERS[key], audio.shape[0] // HOP_LENGTH)[0]


class AudioDataset(Dataset):
    """Items: (audio f32 [480000], decoder_input i64, decoder_output i64, aug i32 [8], extremes i32 [2], cut_frames).

    hu_dataset: indexable records {"audio": {"array"}, "text", "language", optional "prompt"}; tokenizer: whisper-style
    (sot, eot, sot_prev, no_timestamps, no_speech, timestamp_begin, special_tokens[...], encode(text, **kw)).

    Same constructor arguments and attributes as the reference's class (data/data_loader.py:40-160).  `__getitem__`
    ships the raw clip plus the DRAWN augmentation parameters (the batch is turned into mels on the device by
    GpuMelLoader); `_calculate_mel` is the reference's per-clip form of the same arithmetic on the same kernels."""

    def __init__(self, hu_dataset, tokenizer, device=None, no_timestamp_training: bool = False, n_mels: int = 80,
                 max_prompt_length: int = 223, prompt_use_rate: float = 0.5, no_timestamps_rate: float = 0.5,
                 spec_augment: bool = False, spec_augment_params: Optional[dict] = None, extremes_spec_augment: bool = False,
                 extremes_spec_augment_params: Optional[dict] = None, apply_baseline_aug: bool = False, apply_office_aug: bool = False,
                 apply_advanced_aug: bool = False, time_stretch_min_rate: float = 0.8, time_stretch_max_rate: float = 1.25,
                 bpe_dropout: float = 0.0):
        if apply_baseline_aug or apply_office_aug or apply_advanced_aug:
            raise NotImplementedError("audio-domain augmentation (model/augment.py) is outside the GPU hot path (SURVEY.md §2)")
        self.hu_dataset, self.tokenizer, self.n_mels, self.device = hu_dataset, tokenizer, n_mels, device
        self.no_timestamp_training = no_timestamp_training
        self.max_prompt_length, self.prompt_use_rate, self.no_timestamps_rate = max_prompt_length, prompt_use_rate, no_timestamps_rate
        self.spec_augment, self.extremes_spec_augment = spec_augment, extremes_spec_augment
        self.apply_baseline_aug = self.apply_office_aug = self.apply_advanced_aug = False
        self.time_stretch_min_rate, self.time_stretch_max_rate = time_stretch_min_rate, time_stretch_max_rate
        self.bpe_dropout = bpe_dropout
        if spec_augment:
            self.spec_augment_p = float(spec_augment_params.get("p", 1.0))
            if not 0.0 <= self.spec_augment_p <= 1.0:
                raise ValueError(f"spec_augment p must be between 0 and 1, got {self.spec_augment_p}")
            self.time_masking = T.TimeMasking(time_mask_param=spec_augment_params["time_mask_param"])
            self.freq_masking = T.FrequencyMasking(freq_mask_param=spec_augment_params["freq_mask_param"])
            self.time_warping = TimeWarpAugmenter(W=spec_augment_params["time_warp_w"])
        else:
            self.spec_augment_p = 0.0
            self.time_masking = self.freq_masking = self.time_warping = None
        self.extreme_freq_masking = (ExtremesFrequencyMasking(low_freq_range=extremes_spec_augment_params["low_freq_range"],
                                                              high_freq_range=extremes_spec_augment_params["high_freq_range"])
                                     if extremes_spec_augment else None)
        self.aud_augment = None
        self.num_frames_per_second = N_FRAMES / CHUNK_LENGTH
        self.timestamp_pattern = _TS
        self.model_n_text_ctx = 448
        cols = getattr(hu_dataset, "column_names", None)
        if cols is not None:
            assert {"audio", "text", "language"} <= set(cols), "dataset needs audio / text / language columns"
        self.invalid_indices = set()

    # ---- augmentation (reference: data_loader.py:273-301)
    def _should_apply_spec_augment(self) -> bool:
        if not self.spec_augment:
            return False
        if self.spec_augment_p >= 1.0:
            return True
        if self.spec_augment_p <= 0.0:
            return False
        return torch.rand(1).item() < self.spec_augment_p

    def _calculate_mel(self, audio_array, next_partial_segment_start: Optional[float], no_timestamps: bool) -> torch.Tensor:
        """Per-clip form: log-mel -> cut at a partial segment + minimum-value pad -> [warp -> time mask -> freq mask] w.p. p
        -> extremes masking, every stage a libwft kernel on the device."""
        if self.aud_augment is not None:
            audio_array = self.aud_augment(audio_array, sample_rate=16000)
        mel = log_mel_spectrogram(audio_array, n_mels=self.n_mels, device=self.device)
        if no_timestamps and next_partial_segment_start is not None:
            mel = mel[:, : int(next_partial_segment_start * self.num_frames_per_second)]
        if mel.shape[1] != N_FRAMES:
            mel = pad_or_trim(mel, N_FRAMES)
        if self._should_apply_spec_augment():
            mel = self.time_warping(mel)
            mel = self.time_masking(mel)
            mel = self.freq_masking(mel)
        if self.extreme_freq_masking:
            mel = self.extreme_freq_masking(mel)
        return mel

    def _draw_aug_params(self):
        """The draws `_calculate_mel` would make for one clip (same default-generator order), as kernel arguments."""
        return draw_clip_params(self._should_apply_spec_augment,
                                self.time_warping.W if self.time_warping is not None else 0,
                                self.time_masking, self.freq_masking, self.n_mels,
                                self.extreme_freq_masking, N_FRAMES)

    def __len__(self) -> int:
        return len(self.hu_dataset)

    # ---- records
    def _load_valid_record(self, index: int):
        """Corrupt rows are detected lazily and skipped: up to min(len, 32) successors are tried."""
        n = len(self.hu_dataset)
        if n == 0:
            raise IndexError("Dataset is empty.")
        attempts = min(n, 32)
        for off in range(attempts):
            cand = (index + off) % n
            if cand in self.invalid_indices:
                continue
            try:
                rec = self.hu_dataset[cand]
                torch.as_tensor(rec["audio"]["array"])
                if not isinstance(rec["text"], str):
                    raise TypeError(f"Text is not a string: {rec['text']}")
                return cand, rec
            except Exception as exc:
                self.invalid_indices.add(cand)
                print(f"Skipping invalid dataset record at index {cand}: {exc}")
        raise RuntimeError(f"Failed to load a valid record after {attempts} attempts starting from index {index}. "
                           f"Known invalid records so far: {len(self.invalid_indices)}")

    # ---- tokens
    def _encode(self, text: str, keep_timestamps: bool) -> List[int]:
        """`<|t.tt|>` -> timestamp_begin + round(t*100)//2 (t in [0,30], multiple of 0.02) or dropped."""
        out: List[int] = []
        for part in (x for x in _TS.split(text) if x != ""):
            if _TS.fullmatch(part):
                t = float(part[2:-2])
                if t < 0 or t > 30 or round(t * 100) % 2 != 0:
                    raise ValueError(f"Invalid timestamp: {t}")
                if keep_timestamps:
                    out.append(self.tokenizer.timestamp_begin + round(t * 100) // 2)
            else:
                kw = {"dropout_prob": self.bpe_dropout} if self.bpe_dropout else {}
                out.extend(self.tokenizer.encode(part, **kw))
        return out

    def _encode_text_with_timestamps(self, text: str) -> List[int]:
        return self._encode(text, True)

    def _encode_text_without_timestamps(self, text: str) -> List[int]:
        return self._encode(text, False)

    def _get_prompt_tokens(self, record, no_timestamps: bool) -> List[int]:
        prompt = record.get("prompt", "") if hasattr(record, "get") else record["prompt"]
        if torch.rand(1).item() < self.prompt_use_rate and len(prompt) > 0:
            toks = self._encode(prompt, not no_timestamps)[-self.max_prompt_length:]
            return [self.tokenizer.sot_prev] + toks
        return []

    def _get_special_tokens(self, is_text_empty: bool, language: str, no_timestamps: bool) -> List[int]:
        toks = [self.tokenizer.sot, self.tokenizer.special_tokens[f"<|{language}|>"], self.tokenizer.special_tokens["<|transcribe|>"]]
        if no_timestamps:
            toks.append(self.tokenizer.no_timestamps)
        if is_text_empty:
            toks.append(self.tokenizer.no_speech)
        return toks

    def _get_partial_segment_start(self, tokens: List[int]) -> Optional[float]:
        tb = self.tokenizer.timestamp_begin
        if len(tokens) >= 2 and tokens[-2] >= tb and tokens[-1] >= tb:
            return (tokens[-1] - tb) * 0.02
        return None

    def _get_text_tokens(self, text: str, no_timestamps: bool):
        toks = self._encode(text, True)
        start = self._get_partial_segment_start(toks)
        if no_timestamps:
            toks = [t for t in toks if t < self.tokenizer.timestamp_begin]
        return toks, start

    def _construct_decoder_output(self, prompt_tokens, special_tokens, text_tokens) -> List[int]:
        """Targets = inputs shifted by one + eot; the prompt (all but its hand-over to sot) is masked with -100."""
        if not prompt_tokens:
            return special_tokens[1:] + text_tokens + [self.tokenizer.eot]
        return [-100] * (len(prompt_tokens) - 1) + special_tokens + text_tokens + [self.tokenizer.eot]

    def __getitem__(self, index: int):
        index, rec = self._load_valid_record(index)
        no_ts = self.no_timestamp_training or torch.rand(1).item() < self.no_timestamps_rate
        prompt = self._get_prompt_tokens(rec, no_ts)
        text, seg_start = self._get_text_tokens(rec["text"], no_ts)
        special = self._get_special_tokens(len(text) == 0, rec["language"], no_ts)
        dec_in = prompt + special + text
        if len(dec_in) > self.model_n_text_ctx:
            print(f"Input is too long (length: {len(dec_in)}). Shortening... the prompt")
            prompt = prompt[: -(len(dec_in) - self.model_n_text_ctx)]
            dec_in = prompt + special + text
        dec_out = self._construct_decoder_output(prompt, special, text)
        audio = np.asarray(rec["audio"]["array"], dtype=np.float32)
        audio = np.pad(audio, (0, N_SAMPLES - audio.shape[0]), "constant")  # pad in the audio domain (negative pad raises, as upstream)
        cut = int(seg_start * self.num_frames_per_second) if (no_ts and seg_start is not None) else N_FRAMES
        params, ext = self._draw_aug_params()
        return (torch.from_numpy(audio), torch.tensor(dec_in, dtype=torch.int64), torch.tensor(dec_out, dtype=torch.int64),
                params, ext, cut)


def collate_fn(data):
    """x stacked (the reference pads mels with 0), y_in padded with 0, y_out with -100."""
    cols = list(zip(*data))
    y_in = pad_sequence(cols[1], batch_first=True, padding_value=0)
    y_out = pad_sequence(cols[2], batch_first=True, padding_value=-100)
    if len(cols) == 3:
        return pad_sequence(cols[0], batch_first=True, padding_value=0), y_in, y_out
    return torch.stack(cols[0]), y_in, y_out, torch.stack(cols[3]), torch.stack(cols[4]), torch.tensor(cols[5], dtype=torch.int32)


class WarmupDatasetSampler(torch.utils.data.Sampler):
    """Draws only from `warmup_indices` for the first warmup_steps*batch_size samples, then from all indices;
    every pass over an index list is reshuffled with numpy's global RNG (data_loader.py:370-448)."""

    def __init__(self, warmup_indices: Sequence[int], all_indices: Sequence[int], warmup_steps: int, batch_size: int, shuffle: bool = True):
        self.warmup_indices, self.all_indices = list(warmup_indices), list(all_indices)
        if warmup_steps < 0:
            raise ValueError(f"warmup_steps must be >= 0, got {warmup_steps}")
        if batch_size <= 0:
            raise ValueError(f"batch_size must be > 0, got {batch_size}")
        if not self.all_indices:
            raise ValueError("all_indices must be non-empty")
        if not self.warmup_indices and warmup_steps > 0:
            raise ValueError("warmup_indices must be non-empty when warmup_steps > 0")
        self.warmup_steps, self.batch_size, self.shuffle = int(warmup_steps), int(batch_size), shuffle
        self.warmup_samples = self.warmup_steps * self.batch_size

    def __iter__(self) -> Iterator[int]:
        emitted = 0
        while True:
            pool = list(self.warmup_indices if emitted < self.warmup_samples else self.all_indices)
            if self.shuffle:
                np.random.shuffle(pool)
            for idx in pool:
                yield idx
                emitted += 1

    def __len__(self):
        return len(self.all_indices)


def get_dataset_boundary_indices(dataset_sizes: List[int]) -> List[Tuple[int, int]]:
    """[1000, 500, 2000] -> [(0, 1000), (1000, 1500), (1500, 3500)]"""
    out, start = [], 0
    for n in dataset_sizes:
        out.append((start, start + n))
        start += n
    return out


class GpuMelLoader:
    """Wraps the raw-audio DataLoader: pinned, double-buffered H2D of the raw clips (1.92 MB per clip instead of the
    1.54 MB mel plus its CPU cost), then log-mel + SpecAugment on the device.

    With DataLoader workers (num_workers > 0) batch i+1 is fetched and its host-to-device copies are issued on a side
    stream while batch i is being consumed; the compute stream waits on the copy's event only when it starts using the
    batch.  With num_workers == 0 the dataset's augmentation draws share the default generator with the model's own draws
    (stochastic depth, deep SpecAugment): fetching ahead would reorder them against the reference, so that case stays
    strictly sequential.  Yields (mel f32 [B, n_mels, 3000], y_in, y_out), all on the device."""

    def __init__(self, loader, frontend: GpuFrontend, training_aug: bool):
        self.loader, self.frontend, self.training_aug = loader, frontend, training_aug
        self.sampler = getattr(loader, "sampler", None)  # infinite_iter() calls sampler.set_epoch(...)
        self.batch_size = getattr(loader, "batch_size", None)
        self.prefetch = getattr(loader, "num_workers", 0) > 0
        self._copy_stream = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        """Issue the batch's H2D copies on the side stream; returns the device tensors and the event that orders them."""
        dev = self.frontend.device
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        audio, y_in, y_out, params, ext, cut = batch
        need_aug = bool(params[:, 0].any() or ext.any())  # decided on the host copies: no device sync
        with torch.cuda.stream(self._copy_stream):
            staged = [t.to(dev, non_blocking=True) for t in (audio, y_in, y_out, params, ext)]
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return staged, cut, ev, need_aug

    def _to_mel(self, staged, cut, ev, need_aug):
        from whisper_finetune.engine import kernels as K

        cur = torch.cuda.current_stream(self.frontend.device)
        if ev is not None:
            cur.wait_event(ev)
            for t in staged:
                t.record_stream(cur)  # allocated on the copy stream, consumed on the compute stream
        audio, y_in, y_out, params, ext = staged
        mel = self.frontend.log_mel(audio)
        for b in (cut < N_FRAMES).nonzero().flatten().tolist():  # rare: cut at a partial segment, pad with the min value
            c = int(cut[b])
            mel[b, :, c:] = mel[b, :, :c].min() if c > 0 else mel[b].min()
        if self.training_aug and need_aug:
            mel = K.specaug(mel, params, ext)
        return mel, y_in, y_out

    def __iter__(self):
        dev = self.frontend.device
        if not self.prefetch:
            for audio, y_in, y_out, params, ext, cut in self.loader:
                need_aug = bool(params[:, 0].any() or ext.any())
                staged = [t.to(dev, non_blocking=True) for t in (audio, y_in, y_out, params, ext)]
                yield self._to_mel(staged, cut, None, need_aug)
            return
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._stage(next(it))  # copies of batch i+1 run beside the kernels of batch i
            except StopIteration:
                nxt = None
            yield self._to_mel(*cur)


def get_dataloader(hu_dataset, tokenizer, batch_size: int = 1, n_mels: int = 80, sampler=None, device=None,
                   no_timestamp_training: bool = False, max_prompt_length: int = 223, prompt_use_rate: float = 0.5,
                   no_timestamps_rate: float = 0.5, shuffle: bool = True, num_workers: int = 0, spec_augment: bool = False,
                   spec_augment_params: Optional[dict] = None, extremes_spec_augment: bool = False,
                   extremes_spec_augment_params: Optional[dict] = None, apply_baseline_aug: bool = False, apply_office_aug: bool = False,
                   apply_advanced_aug: bool = False, time_stretch_min_rate: float = 0.8, time_stretch_max_rate: float = 1.25,
                   bpe_dropout: float = 0.0, drop_last: bool = False):
    print(f"Found {len(hu_dataset)} records in the dataset.")
    ds = AudioDataset(hu_dataset, tokenizer, n_mels=n_mels, device=device, no_timestamp_training=no_timestamp_training,
                      max_prompt_length=max_prompt_length, prompt_use_rate=prompt_use_rate, no_timestamps_rate=no_timestamps_rate,
                      spec_augment=spec_augment, spec_augment_params=spec_augment_params, extremes_spec_augment=extremes_spec_augment,
                      extremes_spec_augment_params=extremes_spec_augment_params, bpe_dropout=bpe_dropout,
                      apply_baseline_aug=apply_baseline_aug, apply_office_aug=apply_office_aug, apply_advanced_aug=apply_advanced_aug,
                      time_stretch_min_rate=time_stretch_min_rate, time_stretch_max_rate=time_stretch_max_rate)
    if sampler is not None:
        shuffle = False  # DataLoader does not allow both
    loader = DataLoader(ds, batch_size=batch_size, sampler=sampler, shuffle=shuffle, num_workers=num_workers,
                        pin_memory=torch.cuda.is_available(), drop_last=drop_last, collate_fn=collate_fn)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != "cuda":
        return loader  # raw-audio batches (CPU plumbing / tests); the GPU front end needs a device
    fe = GpuFrontend(n_mels, device, spec_augment, spec_augment_params, extremes_spec_augment, extremes_spec_augment_params)
    return GpuMelLoader(loader, fe, training_aug=spec_augment or extremes_spec_augment)


# ---------------------------------------------------------------------------------------------- synthetic provider
class SimpleTokenizer:
    """Whisper-shaped tokenizer for synthetic runs and tests (no tiktoken ranks here): bytes as text tokens and the
    multilingual special-token layout that `get_tokenizer(multilingual=True)` yields with its default 99 languages
    (SURVEY.md finding 6: sot 50258, <|de|> 50261, transcribe 50359, sot_prev 50361, nospeech 50362,
    notimestamps 50363, timestamp_begin 50364, eot 50257)."""

    eot, sot, sot_prev, no_speech, no_timestamps, timestamp_begin = 50257, 50258, 50361, 50362, 50363, 50364

    def __init__(self):
        self.special_tokens = {"<|endoftext|>": 50257, "<|startoftranscript|>": 50258, "<|en|>": 50259, "<|de|>": 50261,
                               "<|translate|>": 50358, "<|transcribe|>": 50359, "<|startofprev|>": 50361,
                               "<|nospeech|>": 50362, "<|notimestamps|>": 50363}

    def encode(self, text: str, **kwargs) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids) -> str:
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="replace")


class SyntheticDataset:
    """HF-datasets-shaped records: N(0, 0.1^2) audio of random length <= 30 s, lower-case pseudo words."""

    column_names = ["audio", "text", "language", "prompt"]

    def __init__(self, n: int, seed: int = 1234, min_seconds: float = 5.0, language: str = "de", with_timestamps: bool = False):
        self.n, self.seed, self.min_seconds, self.language, self.with_timestamps = n, seed, min_seconds, language, with_timestamps

    def __len__(self):
        return self.n

    def __getitem__(self, i: int):
        g = torch.Generator().manual_seed(self.seed + i)
        secs = self.min_seconds + (30.0 - self.min_seconds) * torch.rand(1, generator=g).item()
        audio = torch.randn(int(secs * 16000), generator=g) * 0.1
        words = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(torch.randint(2, 8, (1,), generator=g)),), generator=g))
                 for _ in range(int(torch.randint(3, 20, (1,), generator=g)))]
        text = " ".join(words)
        if self.with_timestamps:
            end = round(min(secs, 30.0) * 50) / 50
            text = f"<|0.00|>{text}<|{end:.2f}|>"
        return {"audio": {"array": audio.numpy(), "sampling_rate": 16000}, "text": text, "language": self.language, "prompt": ""}
