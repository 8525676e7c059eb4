"""`TimeMasking` / `FrequencyMasking` with torchaudio's names and draw order — the namespace the reference imports as
`T` (`import torchaudio.transforms as T`: data/data_loader.py:7,115-116, model/model_utils.py:10,393-394; SURVEY.md
App. A.4).  torchaudio is not a dependency of this build: the span draw is host code (two `torch.rand(1)` on the default
CPU generator: value, then min_value), the zero fill is the `wft_specaug` kernel.

A masker is used three ways:
  * `draw(size)` -> (start, end): the GPU front end and the fused deep-SpecAugment LayerNorm take the drawn span as a
    kernel argument (no separate masking pass at all);
  * `masker(spec)` on a device tensor [..., freq, time]: one `wft_specaug` launch (the reference's per-clip call form);
  * tests / user code may replace the classes (the reference's tests monkeypatch `T.TimeMasking`): callers only rely on
    "construct with the mask parameter, call on a tensor".
"""
from __future__ import annotations

from typing import Tuple

import torch


def draw_mask_span(mask_param: int, size: int) -> Tuple[int, int]:
    """torchaudio `mask_along_axis` draw: value = rand*mask_param; min_value = rand*(size - value);
    [long(min_value), long(min_value) + long(value))."""
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


class _AxisMasking(torch.nn.Module):
    _time_axis = True

    def __init__(self, mask_param: int, iid_masks: bool = False):
        super().__init__()
        if iid_masks:
            raise NotImplementedError("iid_masks=True is not used by the reference and not built")
        self.mask_param = mask_param

    def draw(self, size: int) -> Tuple[int, int]:
        return draw_mask_span(self.mask_param, size)

    def forward(self, specgram: torch.Tensor, mask_value: float = 0.0) -> torch.Tensor:
        """specgram [..., freq, time] on the device; one span for all leading dims (iid_masks=False), filled with 0."""
        if mask_value != 0.0:
            raise NotImplementedError("the reference only masks with 0.0")
        from whisper_finetune.engine import kernels as K
        from whisper_finetune.engine.lib import WftError

        if not specgram.is_cuda:
            raise WftError("TimeMasking / FrequencyMasking run on the GPU in this build (wft_specaug): move the spectrogram to the device")
        shape = specgram.shape
        x = specgram.reshape(-1, shape[-2], shape[-1]).float().contiguous()
        s0, s1 = self.draw(shape[-1] if self._time_axis else shape[-2])
        params = torch.zeros((x.shape[0], 8), dtype=torch.int32)
        if self._time_axis:
            params[:, 3], params[:, 4] = s0, s1
        else:
            params[:, 5], params[:, 6] = s0, s1
        return K.specaug(x, params.to(x.device)).reshape(shape).to(specgram.dtype)


class TimeMasking(_AxisMasking):
    """Zero one span of at most `time_mask_param` steps along the LAST axis."""

    _time_axis = True

    def __init__(self, time_mask_param: int, iid_masks: bool = False, p: float = 1.0):
        super().__init__(time_mask_param, iid_masks)
        if p != 1.0:
            raise NotImplementedError("TimeMasking(p != 1.0) is not used by the reference and not built")


class FrequencyMasking(_AxisMasking):
    """Zero one span of at most `freq_mask_param` bins along the SECOND-TO-LAST axis."""

    _time_axis = False

    def __init__(self, freq_mask_param: int, iid_masks: bool = False):
        super().__init__(freq_mask_param, iid_masks)
