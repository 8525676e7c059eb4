"""GPU front end of the data path: raw 30 s / 16 kHz clips -> log-mel -> SpecAugment, batched on
the device.

It replaces, arithmetic for arithmetic, what `AudioDataset._calculate_mel`
(data/data_loader.py:273-292) does per clip in a CPU DataLoader worker:
  whisper.audio.log_mel_spectrogram -> pad_or_trim -> [time warp -> time mask -> freq mask] w.p. p
  -> extremes frequency masking.
All random parameters are drawn on the HOST from the default CPU generator in the reference's
order (per clip: gate `rand` if 0<p<1; warp `randint(W, L-W)`, `randint(-W, W)`; time mask
`rand, rand`; freq mask `rand, rand`; extremes `rand`) and handed to the kernels, so a seeded
run reproduces the reference's augmentation decisions exactly.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from whisper_finetune.engine import kernels as K

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_SAMPLES = 480000
N_FRAMES = 3000


def mel_filters(n_mels: int) -> torch.Tensor:
    """Slaney-scale, slaney-normalised mel filterbank for sr=16 kHz, n_fft=400 (the asset
    whisper ships as mel_filters.npz): f32 [n_mels, 201]."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), f * 3.0 / 200.0)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)

    n_bins = N_FFT // 2 + 1
    freqs = np.linspace(0, SAMPLE_RATE / 2, n_bins)
    pts = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(SAMPLE_RATE / 2), n_mels + 2))
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        up = (freqs - pts[i]) / (pts[i + 1] - pts[i])
        down = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
        w[i] = np.maximum(0.0, np.minimum(up, down))
    w *= (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


from whisper_finetune.data.transforms import FrequencyMasking, TimeMasking, draw_mask_span  # noqa: E402,F401


def draw_clip_params(should_apply, warp_w: int, time_masking, freq_masking, n_mels: int, extremes, T: int = N_FRAMES):
    """Host draws of ONE clip in the order `AudioDataset._calculate_mel` makes them (data/data_loader.py:284-290):
    gate -> warp `randint(W, T-W)`, `randint(-W, W)` (data/utils.py:107,111) -> time span -> frequency span -> extremes
    ratio.  Returns (params i32 [8] = {apply_warp, warp_p, warp_d, t0, t1, f0, f1, 0}, extremes i32 [2]) for wft_specaug.
    `should_apply` is a callable (the dataset's `_should_apply_spec_augment`); the maskers offer `draw(size)`;
    `extremes` is an ExtremesFrequencyMasking or None."""
    params = np.zeros(8, dtype=np.int32)
    ext = np.zeros(2, dtype=np.int32)
    if should_apply():
        warp_p = int(torch.randint(warp_w, T - warp_w, (1,)))
        warp_d = int(torch.randint(-warp_w, warp_w, (1,)))
        t0, t1 = time_masking.draw(T)
        f0, f1 = freq_masking.draw(n_mels)
        params[:] = (1, warp_p, warp_d, t0, t1, f0, f1, 0)
    if extremes is not None:
        r = torch.rand(1).item()
        ext[:] = (int(round(r * extremes.low_freq_range)), int(round(r * extremes.high_freq_range)))
    return torch.from_numpy(params), torch.from_numpy(ext)


class GpuFrontend:
    def __init__(self, n_mels: int, device, spec_augment: bool = False, spec_augment_params: Optional[dict] = None,
                 extremes_spec_augment: bool = False, extremes_spec_augment_params: Optional[dict] = None):
        self.n_mels = n_mels
        self.device = torch.device(device)
        self.filters = mel_filters(n_mels).to(self.device)
        self.spec_augment = spec_augment
        p = spec_augment_params or {}
        self.p = float(p.get("p", 1.0)) if spec_augment else 0.0
        if spec_augment and not 0.0 <= self.p <= 1.0:
            raise ValueError(f"spec_augment p must be between 0 and 1, got {self.p}")
        self.time_mask_param = p.get("time_mask_param", 0)
        self.freq_mask_param = p.get("freq_mask_param", 0)
        self.time_warp_w = p.get("time_warp_w", 0)
        e = extremes_spec_augment_params or {}
        self.extremes = extremes_spec_augment
        self.low_freq_range = e.get("low_freq_range", 0)
        self.high_freq_range = e.get("high_freq_range", 0)

    def _should_apply(self) -> bool:
        if not self.spec_augment or self.p <= 0.0:
            return False
        return True if self.p >= 1.0 else torch.rand(1).item() < self.p

    def draw(self, batch: int, T: int = N_FRAMES):
        """Host draws for a batch -> (params i32 [B,8], extremes i32 [B,2]) in the reference's per-clip order."""
        from whisper_finetune.data.utils import ExtremesFrequencyMasking

        tm, fm = TimeMasking(self.time_mask_param), FrequencyMasking(self.freq_mask_param)
        ex = ExtremesFrequencyMasking(self.low_freq_range, self.high_freq_range) if self.extremes else None
        rows = [draw_clip_params(self._should_apply, self.time_warp_w, tm, fm, self.n_mels, ex, T) for _ in range(batch)]
        return torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])

    def log_mel(self, audio: torch.Tensor) -> torch.Tensor:
        """audio f32 [B, 480000] on the device -> f32 [B, n_mels, 3000]."""
        return K.logmel(audio, self.filters, audio.shape[1] // HOP_LENGTH)

    def __call__(self, audio: torch.Tensor, training: bool = True) -> torch.Tensor:
        mel = self.log_mel(audio)
        if training and (self.spec_augment or self.extremes):
            params, ext = self.draw(audio.shape[0], mel.shape[-1])
            if params[:, 0].any() or ext.any():
                mel = K.specaug(mel, params.to(self.device, non_blocking=True), ext.to(self.device, non_blocking=True))
        return mel
