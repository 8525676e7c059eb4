"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain torch-fp32 (CPU) restatement of the arithmetic on the reference's training hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (whisper-finetune_amd/) never does.

Paths cited below are relative to the reference root (i4Ds/whisper-finetune).  The model
arithmetic itself lives in third-party packages that are NOT vendored in the reference and
not installed here: openai-whisper >= 20240930 (pyproject.toml:12), minLoRA git main
(pyproject.toml:23), torchaudio >= 2.1.2 (pyproject.toml:16).  Their published algorithms
are restated from SURVEY.md Appendix A, anchored on the reference's own call sites.

Parity pinning (tests/golden/, generator scripts committed next to the fixtures):
  * model forward / loss / gradients: an independent implementation of the same
    architecture (HF transformers WhisperForConditionalGeneration, bridged with the
    reference's key map scripts/convert_openai_to_hf.py:89-110) -> whisper_arch.npz
  * log-mel: HF WhisperFeatureExtractor (numpy STFT path) -> logmel.npz
  * time warp / extremes masking / pad_or_trim / stochastic depth / deep-SpecAugment hooks /
    train_step control flow: the REFERENCE'S OWN Python imported in this container with
    import stubs for the missing third-party packages -> ref_host.npz
  * LoRA algebra: the invariants the reference's tests pin (tests/test_lora.py:42-44,
    124-127,262-273,612-614,762-766), restated in tests/test_oracle.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------------------------
# constants (whisper.audio; mirrored by the reference's stub tests/test_data_loader.py:27-31 and
# whisper_v3_utils/preprocessor_config.json:2-12)
SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH  # 3000


@dataclass
class ModelDimensions:
    """whisper.model.ModelDimensions (fields read at scripts/finetune.py:433-447)."""

    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


DIMS = {
    # SURVEY.md App. A.1 / whisper_v3_utils/config.json:16-26,37-49
    "tiny": ModelDimensions(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base": ModelDimensions(80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small": ModelDimensions(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium": ModelDimensions(80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v3": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    "large-v3-turbo": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
}


# ----------------------------------------------------------------------------------------------
# A1 — log-mel (data/data_loader.py:278 -> whisper.audio.log_mel_spectrogram)
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filters(n_mels: int) -> Tensor:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels) (slaney scale + slaney norm) = the
    whisper/assets/mel_filters.npz asset; f32 [n_mels, 201]."""
    n_bins = N_FFT // 2 + 1
    fft_freqs = np.linspace(0, SAMPLE_RATE / 2, n_bins)
    mel_pts = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(SAMPLE_RATE / 2), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fft_freqs[None, :]
    weights = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2 : n_mels + 2] - mel_pts[:n_mels])
    weights *= enorm[:, None]
    return torch.from_numpy(weights.astype(np.float32))


def log_mel_spectrogram(audio: Tensor, n_mels: int = 80) -> Tensor:
    """f32 [..., n_samples] -> f32 [..., n_mels, n_samples // 160]; the max is per clip."""
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = mel_filters(n_mels) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    if log_spec.dim() == 2:
        mx = log_spec.max()
    else:
        mx = log_spec.amax(dim=(-2, -1), keepdim=True)
    log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


# ----------------------------------------------------------------------------------------------
# A2 — SpecAugment (data/data_loader.py:284-290; data/utils.py:41-143,146-190,380-404)
def time_warp(spec: Tensor, warp_p: int, warp_d: int) -> Tensor:
    """TimeWarpAugmenter.time_warp for ONE spectrogram [n_mels, L] with already-drawn
    warp_p in [W, L-W) and warp_d in [-W, W) (data/utils.py:107,111): cubic Hermite
    through (0,-1), (warp_p, y1), (L-1, 1) then bilinear grid_sample(align_corners=True)."""
    num_rows, spec_len = spec.shape
    # integer control points and fp32 arithmetic exactly as the reference forms them (data/utils.py:113-136)
    wp = torch.tensor([warp_p], dtype=torch.int64)
    wd = torch.tensor([warp_d], dtype=torch.int64)
    x = torch.stack([torch.tensor([0]), wp, torch.tensor([spec_len - 1])], 1)  # int64 [1, 3]
    y = torch.stack([torch.tensor([-1.0]), (wp - wd) * 2 / (spec_len - 1.0) - 1.0, torch.tensor([1.0])], 1)
    xs = torch.linspace(0, spec_len - 1, spec_len).unsqueeze(0)
    # hspline_interpolate_1D (data/utils.py:66-85)
    m = (y[..., 1:] - y[..., :-1]) / (x[..., 1:] - x[..., :-1])
    m = torch.cat([m[..., [0]], (m[..., 1:] + m[..., :-1]) / 2, m[..., [-1]]], -1)
    idxs = torch.searchsorted(x[..., 1:], xs)
    dx = x.gather(dim=-1, index=idxs + 1) - x.gather(dim=-1, index=idxs)
    t = (xs - x.gather(dim=-1, index=idxs)) / dx
    tt = t.unsqueeze(-2) ** torch.arange(4).view(-1, 1)
    A = torch.tensor([[1, 0, -3, 2], [0, 1, -2, 1], [0, 0, 3, -2], [0, 0, -1, 1]], dtype=t.dtype)
    hh = A @ tt
    ys = (
        hh[..., 0, :] * y.gather(dim=-1, index=idxs)
        + hh[..., 1, :] * m.gather(dim=-1, index=idxs) * dx
        + hh[..., 2, :] * y.gather(dim=-1, index=idxs + 1)
        + hh[..., 3, :] * m.gather(dim=-1, index=idxs + 1) * dx
    )
    grid = torch.cat(
        (
            ys.view(1, 1, -1, 1).expand(-1, num_rows, -1, -1),
            torch.linspace(-1, 1, num_rows).view(-1, 1, 1).expand(1, -1, spec_len, -1),
        ),
        -1,
    )
    return F.grid_sample(spec[None, None], grid, align_corners=True)[0, 0]


def draw_mask_span(mask_param: int, size: int, generator=None) -> Tuple[int, int]:
    """torchaudio.functional.mask_along_axis draw order: value then min_value,
    both torch.rand(1) on the default CPU generator; returns [start, end)."""
    value = torch.rand(1, generator=generator) * mask_param
    min_value = torch.rand(1, generator=generator) * (size - value)
    start = int(min_value.long())
    end = start + int(value.long())
    return start, end


def spec_augment(mel: Tensor, warp: Optional[Tuple[int, int]], t_span: Tuple[int, int], f_span: Tuple[int, int],
                 extremes: Tuple[int, int] = (0, 0)) -> Tensor:
    """AudioDataset._calculate_mel tail (data/data_loader.py:284-290): warp -> time mask ->
    freq mask (mask value 0.0) -> extremes masking.  mel [n_mels, T]."""
    out = mel.clone()
    if warp is not None:
        out = time_warp(out, warp[0], warp[1])
    out[:, t_span[0] : t_span[1]] = 0.0
    out[f_span[0] : f_span[1], :] = 0.0
    n_mels = out.shape[0]
    lo, hi = extremes
    if lo > 0:
        out[: min(lo, n_mels)] = 0.0
    if hi > 0:
        out[max(n_mels - hi, 0) :] = 0.0
    return out


def extremes_lengths(r: float, low_freq_range: int, high_freq_range: int) -> Tuple[int, int]:
    """ExtremesFrequencyMasking: one ratio r per sample (data/utils.py:176-186)."""
    return int(round(r * low_freq_range)), int(round(r * high_freq_range))


def pad_or_trim_min(mel: Tensor, length: int = N_FRAMES) -> Tensor:
    """pad_or_trim with the array minimum as pad value (data/utils.py:380-394)."""
    if mel.shape[-1] > length:
        mel = mel[..., :length]
    if mel.shape[-1] < length:
        mel = F.pad(mel, (0, length - mel.shape[-1]), value=torch.min(mel).item())
    return mel


def collate(xs: Sequence[Tensor], y_ins: Sequence[Tensor], y_outs: Sequence[Tensor]):
    """collate_fn (data/data_loader.py:362-367): x pad 0, y_in pad 0, y_out pad -100."""
    from torch.nn.utils.rnn import pad_sequence

    return (
        pad_sequence(list(xs), batch_first=True, padding_value=0),
        pad_sequence(list(y_ins), batch_first=True, padding_value=0),
        pad_sequence(list(y_outs), batch_first=True, padding_value=-100),
    )


# ----------------------------------------------------------------------------------------------
# E1/E2/D1/D2 — the model (whisper.model; in-tree copies model/model_utils.py:271-327)
def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> Tensor:
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    scaled = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)


def init_params(dims: ModelDimensions, seed: int = 0, std: float = 0.02, device=None) -> Dict[str, Tensor]:
    """Random-init state dict with openai-whisper key names (scripts/convert_openai_to_hf.py:89-110
    is the in-tree spec of the names).  SURVEY §8d: matrices/embeddings N(0, 0.02^2), biases 0,
    LN gamma 1 beta 0; encoder positions = sinusoids (a buffer, not trained).  `device`: where the normal draws are made
    (default: the CPU generator every fixture was produced with; the large-v3 tests draw their 1.5e9 values on the
    accelerator — 1 s instead of 19 — and get CPU tensors back, a different but equally seeded sample)."""
    g = torch.Generator().manual_seed(seed) if device is None else torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def mat(*shape):
        if device is None:
            return torch.randn(*shape, generator=g) * std
        return (torch.randn(*shape, generator=g, device=device) * std).cpu()

    d = dims.n_audio_state
    sd["encoder.conv1.weight"] = mat(d, dims.n_mels, 3)
    sd["encoder.conv1.bias"] = torch.zeros(d)
    sd["encoder.conv2.weight"] = mat(d, d, 3)
    sd["encoder.conv2.bias"] = torch.zeros(d)
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d)

    def block(prefix: str, d: int, cross: bool):
        names = ["attn"] + (["cross_attn"] if cross else [])
        for a in names:
            sd[f"{prefix}.{a}.query.weight"] = mat(d, d)
            sd[f"{prefix}.{a}.query.bias"] = torch.zeros(d)
            sd[f"{prefix}.{a}.key.weight"] = mat(d, d)  # no bias
            sd[f"{prefix}.{a}.value.weight"] = mat(d, d)
            sd[f"{prefix}.{a}.value.bias"] = torch.zeros(d)
            sd[f"{prefix}.{a}.out.weight"] = mat(d, d)
            sd[f"{prefix}.{a}.out.bias"] = torch.zeros(d)
            sd[f"{prefix}.{a}_ln.weight"] = torch.ones(d)
            sd[f"{prefix}.{a}_ln.bias"] = torch.zeros(d)
        sd[f"{prefix}.mlp.0.weight"] = mat(4 * d, d)
        sd[f"{prefix}.mlp.0.bias"] = torch.zeros(4 * d)
        sd[f"{prefix}.mlp.2.weight"] = mat(d, 4 * d)
        sd[f"{prefix}.mlp.2.bias"] = torch.zeros(d)
        sd[f"{prefix}.mlp_ln.weight"] = torch.ones(d)
        sd[f"{prefix}.mlp_ln.bias"] = torch.zeros(d)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, False)
    sd["encoder.ln_post.weight"] = torch.ones(d)
    sd["encoder.ln_post.bias"] = torch.zeros(d)
    dt = dims.n_text_state
    sd["decoder.token_embedding.weight"] = mat(dims.n_vocab, dt)
    sd["decoder.positional_embedding"] = mat(dims.n_text_ctx, dt)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, True)
    sd["decoder.ln.weight"] = torch.ones(dt)
    sd["decoder.ln.bias"] = torch.zeros(dt)
    return sd


def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """whisper.model.LayerNorm: F.layer_norm(x.float()).type(x.dtype), eps 1e-5."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5).type(x.dtype)


def lora_effective_weight(W: Tensor, A: Tensor, B: Tensor, scaling: float, mask: Optional[Tensor] = None) -> Tensor:
    """minLoRA LoRAParametrization.forward (applied at model/lora.py:54-57):
    W + (B @ (A * mask)).view(W.shape) * scaling;  A [r, in], B [out, r], mask [1, in]."""
    a = A if mask is None else A * mask
    return W + (B @ a).view(W.shape) * scaling


def lora_init(fan_out: int, fan_in: int, rank: int, generator=None) -> Tuple[Tensor, Tensor]:
    """from_linear: A = kaiming_uniform_(a=sqrt(5)) on zeros(rank, fan_in); B = zeros(fan_out, rank)."""
    bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))  # gain(leaky_relu, sqrt5)=sqrt(2/6); bound=gain*sqrt(3/fan_in)
    A = (torch.rand(rank, fan_in, generator=generator) * 2 - 1) * bound
    return A, torch.zeros(fan_out, rank)



# ----------------------------------------------------------------------------------------------
# bf16-emulation mode (tests only): the fp32 restatement above is the parity reference; the emulation rounds to bf16
# at the points where the libwft kernels round (GEMM outputs after their fused epilogue, LayerNorm outputs, the
# probabilities / dS that feed the attention MFMAs, the residual stream), with fp32 accumulation everywhere — the
# arithmetic contract of the reference's autocast(bf16) path (SURVEY.md §7 "Numerics contract").  Comparing the GPU path
# with THIS removes the common rounding noise, so a mis-scaled term in a single tensor shows (per-tensor bound 2e-2
# instead of the 6-8e-2 a bf16-vs-fp32 comparison needs).
def _bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundAct(torch.autograd.Function):
    """Activation rounding: bf16 forward, and the gradient that flows back is a bf16 tensor in HBM too."""

    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _RoundWeight(torch.autograd.Function):
    """Weight shadow rounding: bf16 forward; weight gradients are fp32 GEMM outputs (no rounding)."""

    @staticmethod
    def forward(ctx, w):
        return _bf16(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _EmulatedAttention(torch.autograd.Function):
    """The attention kernels' arithmetic (csrc/attn.hip) on [B, H, T, 64] fp32 tensors holding bf16 values:
    forward  S = q k^T (fp32) -> P = exp(scale*S - max) -> l = sum P (fp32, unrounded P) -> O = (bf16(P) v) / l -> bf16;
    backward recompute P = exp(scale*S - lse), delta = sum_d dO*O, dP = dO v^T, dS = P (dP - delta),
             dV = bf16(P)^T dO, dQ = scale * bf16(dS) k, dK = scale * bf16(dS)^T q, each stored as bf16."""

    @staticmethod
    def forward(ctx, q, k, v, causal: bool, scale: float):
        s = (q @ k.transpose(-1, -2)) * scale
        if causal:
            T = s.shape[-1]
            s = s + torch.full((T, T), float("-inf")).triu_(1)
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = _bf16((_bf16(p) @ v) / l)
        ctx.save_for_backward(q, k, v, o, m + torch.log(l))
        ctx.cfg = (causal, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        causal, scale = ctx.cfg
        do = _bf16(do)
        s = (q @ k.transpose(-1, -2)) * scale
        if causal:
            T = s.shape[-1]
            s = s + torch.full((T, T), float("-inf")).triu_(1)
        p = torch.exp(s - lse)
        delta = (do * o).sum(dim=-1, keepdim=True)
        ds = _bf16(p * (do @ v.transpose(-1, -2) - delta))
        dv = _bf16(_bf16(p).transpose(-1, -2) @ do)
        dq = _bf16((ds @ k) * scale)
        dk = _bf16((ds.transpose(-1, -2) @ q) * scale)
        return dq, dk, dv, None, None


class Oracle:
    """Functional Whisper forward over a state dict (fp32).  Hooks for the reference's training-
    time extras: stochastic depth decisions, deep-SpecAugment masks and LoRA adapters."""

    def __init__(self, dims: ModelDimensions, params: Dict[str, Tensor], lora: Optional[dict] = None, emulate_bf16: bool = False):
        self.dims = dims
        self.p = params
        # lora: {"<module prefix>": (A, B, scaling, mask or None)} keyed by e.g. "encoder.blocks.0.attn.query"
        self.lora = lora or {}
        # emulate_bf16: round where the libwft kernels round (see _RoundAct above); False = the fp32 restatement
        self.emulate = emulate_bf16
        self.ra = _RoundAct.apply if emulate_bf16 else (lambda t: t)
        self.rw = _RoundWeight.apply if emulate_bf16 else (lambda t: t)

    def weight(self, prefix: str) -> Tensor:
        W = self.p[prefix + ".weight"]
        if prefix in self.lora:
            A, B, s, m = self.lora[prefix]
            W = lora_effective_weight(W, A, B, s, m)
        return self.rw(W)

    def linear(self, x: Tensor, prefix: str) -> Tensor:
        return F.linear(x, self.weight(prefix), self.p.get(prefix + ".bias"))

    def attention(self, x: Tensor, prefix: str, n_head: int, xa: Optional[Tensor] = None, causal: bool = False) -> Tensor:
        """MultiHeadAttention.forward + qkv_attention (non-SDPA branch): q,k scaled by d_h^-0.25 each,
        softmax in fp32; mask = -inf strictly above the diagonal (decoder self-attention only)."""
        q = self.ra(self.linear(x, prefix + ".query"))
        src = x if xa is None else xa
        k = self.ra(self.linear(src, prefix + ".key"))
        v = self.ra(self.linear(src, prefix + ".value"))
        B, T, D = q.shape
        if self.emulate:
            def heads(t):
                return t.view(B, t.shape[1], n_head, -1).permute(0, 2, 1, 3)
            o = _EmulatedAttention.apply(heads(q), heads(k), heads(v), causal, (D // n_head) ** -0.5)
            return self.linear(o.permute(0, 2, 1, 3).flatten(start_dim=2), prefix + ".out")
        scale = (D // n_head) ** -0.25
        qh = q.view(B, T, n_head, -1).permute(0, 2, 1, 3) * scale
        kh = k.view(B, k.shape[1], n_head, -1).permute(0, 2, 3, 1) * scale
        vh = v.view(B, v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        qk = qh @ kh
        if causal:
            qk = qk + torch.full((T, T), float("-inf")).triu_(1)
        w = F.softmax(qk.float(), dim=-1).to(q.dtype)
        wv = (w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2)
        return self.linear(wv, prefix + ".out")

    def block(self, x: Tensor, prefix: str, n_head: int, xa: Optional[Tensor] = None, causal: bool = False,
              ln_mask: Optional[Tuple[int, int, int, int]] = None) -> Tensor:
        """ResidualAttentionBlock.  ln_mask=(t0,t1,c0,c1): the deep-SpecAugment forward hook on attn_ln
        (model/model_utils.py:409-417): time rows then channel columns zero-filled."""
        ra = self.ra  # identity in fp32 mode; the residual add is fused in the out-projection / fc2 GEMM epilogues
        h = ra(layer_norm(x, self.p[prefix + ".attn_ln.weight"], self.p[prefix + ".attn_ln.bias"]))
        if ln_mask is not None:
            t0, t1, c0, c1 = ln_mask
            h = h.clone()
            h[:, t0:t1, :] = 0.0
            h[:, :, c0:c1] = 0.0
        x = ra(x + self.attention(h, prefix + ".attn", n_head, causal=causal))
        if xa is not None:
            h = ra(layer_norm(x, self.p[prefix + ".cross_attn_ln.weight"], self.p[prefix + ".cross_attn_ln.bias"]))
            x = ra(x + self.attention(h, prefix + ".cross_attn", n_head, xa=xa))
        h = ra(layer_norm(x, self.p[prefix + ".mlp_ln.weight"], self.p[prefix + ".mlp_ln.bias"]))
        h = ra(F.gelu(self.linear(h, prefix + ".mlp.0")))
        return ra(x + self.linear(h, prefix + ".mlp.2"))

    @staticmethod
    def stochastic_depth(x: Tensor, fn, p: float, training: bool, skip: bool, ra=None) -> Tensor:
        """StochasticDepthMixin.stochastic_depth (model/model_utils.py:226-250) with the
        `torch.rand(1).item() < p` decision passed in as `skip`."""
        if training and p > 0.0 and skip:
            return x
        out = fn(x)
        if training and p > 0.0:
            keep = 1.0 - p
            if keep <= 0.0:
                return x
            y = x + (out - x) / keep
            return y if ra is None else ra(y)  # ra: the bf16 rounding of the fused rescale kernel (emulation mode)
        return out

    def encoder(self, mel: Tensor, sd_p: float = 0.0, training: bool = False, skips: Optional[List[bool]] = None,
                ln_masks: Optional[Dict[int, Tuple[int, int, int, int]]] = None) -> Tensor:
        """model/model_utils.py:271-288."""
        p = self.p
        ra, rw = self.ra, self.rw
        x = ra(F.gelu(F.conv1d(ra(mel), rw(p["encoder.conv1.weight"]), p["encoder.conv1.bias"], padding=1)))
        x = F.gelu(F.conv1d(x, rw(p["encoder.conv2.weight"]), p["encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == p["encoder.positional_embedding"].shape, "incorrect audio shape"
        x = ra((x + rw(p["encoder.positional_embedding"])).to(x.dtype))
        for i in range(self.dims.n_audio_layer):
            mk = (ln_masks or {}).get(i)
            fn = lambda t, i=i, mk=mk: self.block(t, f"encoder.blocks.{i}", self.dims.n_audio_head, ln_mask=mk)  # noqa: E731
            x = self.stochastic_depth(x, fn, sd_p, training, bool(skips[i]) if skips else False, ra=self.ra)
        return self.ra(layer_norm(x, p["encoder.ln_post.weight"], p["encoder.ln_post.bias"]))

    def decoder(self, tokens: Tensor, xa: Tensor, sd_p: float = 0.0, training: bool = False,
                skips: Optional[List[bool]] = None) -> Tensor:
        """model/model_utils.py:309-327 (kv_cache=None in training)."""
        p = self.p
        x = F.embedding(tokens, p["decoder.token_embedding.weight"]) + p["decoder.positional_embedding"][: tokens.shape[-1]]
        x = self.ra(x.to(xa.dtype))
        for i in range(self.dims.n_text_layer):
            fn = lambda t, i=i: self.block(t, f"decoder.blocks.{i}", self.dims.n_text_head, xa=xa, causal=True)  # noqa: E731
            x = self.stochastic_depth(x, fn, sd_p, training, bool(skips[i]) if skips else False, ra=self.ra)
        x = self.ra(layer_norm(x, p["decoder.ln.weight"], p["decoder.ln.bias"]))
        return self.ra(x @ self.rw(p["decoder.token_embedding.weight"]).to(x.dtype).T).float()

    def forward(self, mel: Tensor, tokens: Tensor, **kw) -> Tensor:
        """Whisper.forward = decoder(tokens, encoder(mel)) -> logits f32 [B, S, V]."""
        enc_kw = {k[4:]: v for k, v in kw.items() if k.startswith("enc_")}
        dec_kw = {k[4:]: v for k, v in kw.items() if k.startswith("dec_")}
        return self.decoder(tokens, self.encoder(mel, **enc_kw), **dec_kw)


# ----------------------------------------------------------------------------------------------
# X — loss (model/model_utils.py:66-68)
def cross_entropy(logits: Tensor, y_out: Tensor, label_smoothing: float = 0.0) -> Tensor:
    return F.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing=label_smoothing)


def cross_entropy_manual(logits: Tensor, y_out: Tensor, eps: float = 0.0) -> Tensor:
    """Closed form used by the fused kernel: mean over non-ignored targets of
    (1-eps)*nll + eps*(lse - mean_c logit_c)   (SURVEY §8a row X)."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V).float()
    t = y_out.reshape(-1)
    valid = t != -100
    lse = torch.logsumexp(lg, -1)
    xt = lg.gather(1, t.clamp(min=0).unsqueeze(1)).squeeze(1)
    row = (1 - eps) * (lse - xt) + eps * (lse - lg.mean(-1))
    return (row * valid).sum() / valid.sum()


# V — teacher-forced eval reductions (eval/evaluator.py:70-73, eval/metrics.py:106-137)
def teacher_forced_argmax(logits: Tensor) -> Tensor:
    return logits.argmax(dim=-1)


# ----------------------------------------------------------------------------------------------
# T — train_step arithmetic (model/model_utils.py:54-73): sum over micro-batches of loss/accum
def train_step_loss(oracle: Oracle, batches, accum: int, label_smoothing: float = 0.0) -> Tensor:
    total = 0.0
    for mel, y_in, y_out in batches:
        total = total + cross_entropy(oracle.forward(mel, y_in), y_out, label_smoothing) / accum
    return total


def synthetic_batch(dims: ModelDimensions, B: int, S: int, seed: int = 1234, n_samples: int = N_SAMPLES):
    """SURVEY §8d synthetic inputs: N(0, 0.1^2) audio, random text tokens after the v2-layout
    special tokens [sot, <|de|>, <|transcribe|>, <|notimestamps|>]; y_out = shift + eot."""
    audio = torch.stack([torch.randn(n_samples, generator=torch.Generator().manual_seed(seed + i)) * 0.1 for i in range(B)])
    g = torch.Generator().manual_seed(4321 + seed)
    specials = torch.tensor([50258, 50261, 50359, 50363])
    body = torch.randint(0, 50257, (B, S - 4), generator=g)
    y_in = torch.cat([specials.expand(B, -1), body], dim=1)
    y_out = torch.cat([y_in[:, 1:], torch.full((B, 1), 50257)], dim=1)
    return audio, y_in, y_out


# ----------------------------------------------------------------------------------------------
# §8f-1 — optimizer step.  The reference builds `muon.MuonWithAuxAdam` / `SingleDeviceMuonWithAuxAdam`
# (model/optimizer.py:171-237) from the third-party package `muon` (github.com/KellerJordan/Muon, git HEAD, pinned by
# nothing: pyproject.toml:29) — NOT present in /root/reference and not installed, so its published algorithm is
# restated here (PARITY UNPINNED for this part: no golden vectors exist in the reference's tests beyond the
# param-group keys, tests/test_optimizer.py:22-60).  torch.optim.AdamW is the oracle for the AdamW path.
NS_COEFFS = (3.4445, -4.7750, 2.0315)


def zeropower_via_newtonschulz5(G: Tensor, steps: int = 5, dtype=torch.bfloat16) -> Tensor:
    """Quintic Newton-Schulz iteration towards the orthogonal polar factor of G, in bf16 (muon.py).  `dtype=float32`
    evaluates the same iteration in fp32: tests use the distance between the two as the update's measured sensitivity to
    bf16 rounding (weak singular directions are amplified ~3.4x per iteration, rounding noise with them)."""
    assert G.ndim >= 2
    a, b, c = NS_COEFFS
    X = G.to(dtype)
    if G.size(-2) > G.size(-1):
        X = X.mT
    X = X / (X.norm(dim=(-2, -1), keepdim=True) + 1e-7)
    for _ in range(steps):
        A = X @ X.mT
        B = b * A + c * A @ A
        X = a * X + B @ X
    if G.size(-2) > G.size(-1):
        X = X.mT
    return X


def muon_update(grad: Tensor, momentum: Tensor, beta: float = 0.95, ns_steps: int = 5, nesterov: bool = True,
                ns_dtype=torch.bfloat16) -> Tensor:
    """In place on grad and momentum, as the package does (muon.py muon_update)."""
    momentum.lerp_(grad, 1 - beta)
    update = grad.lerp_(momentum, beta) if nesterov else momentum
    if update.ndim == 4:
        update = update.view(len(update), -1)
    update = zeropower_via_newtonschulz5(update, steps=ns_steps, dtype=ns_dtype)
    update *= max(1, grad.size(-2) / grad.size(-1)) ** 0.5
    return update


def adam_update(grad: Tensor, buf1: Tensor, buf2: Tensor, step: int, betas, eps: float) -> Tensor:
    buf1.lerp_(grad, 1 - betas[0])
    buf2.lerp_(grad.square(), 1 - betas[1])
    buf1c = buf1 / (1 - betas[0] ** step)
    buf2c = buf2 / (1 - betas[1] ** step)
    return buf1c / (buf2c.sqrt() + eps)


def muon_with_aux_adam_step(param_groups, state: dict, ns_dtype=torch.bfloat16) -> None:
    """One SingleDeviceMuonWithAuxAdam.step() over plain tensors: param_groups are the reference's dicts (use_muon, lr,
    momentum | betas/eps, weight_decay) whose "params" are (p, grad) pairs; `state` maps id(p) -> dict."""
    with torch.no_grad():
        for group in param_groups:
            for p, g in group["params"]:
                st = state.setdefault(id(p), {})
                if group["use_muon"]:
                    if g is None:
                        g = torch.zeros_like(p)
                    if not st:
                        st["momentum_buffer"] = torch.zeros_like(p)
                    update = muon_update(g, st["momentum_buffer"], beta=group["momentum"], ns_dtype=ns_dtype)
                    p.mul_(1 - group["lr"] * group["weight_decay"])
                    p.add_(update.reshape(p.shape).to(p.dtype), alpha=-group["lr"])
                else:
                    if g is None:
                        g = torch.zeros_like(p)
                    if not st:
                        st["exp_avg"], st["exp_avg_sq"], st["step"] = torch.zeros_like(p), torch.zeros_like(p), 0
                    st["step"] += 1
                    update = adam_update(g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["betas"], group["eps"])
                    p.mul_(1 - group["lr"] * group["weight_decay"])
                    p.add_(update, alpha=-group["lr"])
