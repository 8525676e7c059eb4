"""Whisper model with openai-whisper-compatible module / attribute / state-dict names
(SURVEY.md §8 b1), whose compute runs on libwft HIP kernels (engine/ops.py).

Mirrors `whisper.model` (absent from the reference tree; restated in SURVEY.md App. A.1 and
by the in-tree copies model/model_utils.py:271-327 and the key map
scripts/convert_openai_to_hf.py:89-110):
  Whisper(dims).encoder: conv1, conv2, positional_embedding (buffer), blocks[i].{attn, attn_ln,
  mlp, mlp_ln}, ln_post;  .decoder: token_embedding, positional_embedding (Parameter), blocks[i]
  .{attn, attn_ln, cross_attn, cross_attn_ln, mlp, mlp_ln}, ln, mask (non-persistent buffer).

Numerics: fp32 master parameters, bf16 activations / MFMA inputs, fp32 accumulation, fp32
LayerNorm statistics — the contract of the reference's autocast(bf16) path.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass
from typing import Dict, Iterable, Optional

import os

import numpy as np

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import kernels as K
from . import ops
from . import ops32

BF16 = torch.bfloat16


# WFT_XA_ACCUM=0: the encoder-output gradient is summed by autograd (A/B runs)
_XA_ACCUM = os.environ.get("WFT_XA_ACCUM", "1") != "0"


@dataclass
class ModelDimensions:
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


MODEL_DIMS: Dict[str, ModelDimensions] = {
    "tiny": ModelDimensions(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base": ModelDimensions(80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small": ModelDimensions(80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium": ModelDimensions(80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v2": ModelDimensions(80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32),
    "large-v3": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    "large-v3-turbo": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
}
MODEL_DIMS["large"] = MODEL_DIMS["large-v3"]
MODEL_DIMS["turbo"] = MODEL_DIMS["large-v3-turbo"]


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> Tensor:
    assert channels % 2 == 0
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    scaled = torch.arange(length)[:, np.newaxis] * inv[np.newaxis, :]
    return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)


def _as2d(x: Tensor) -> Tensor:
    return x.reshape(-1, x.shape[-1])


def _to_bf16(x: Tensor) -> Tensor:
    if x.dtype == BF16:
        return x.contiguous()
    if x.dtype == torch.float32 and not x.requires_grad:
        return K.cast_bf16(x)
    return x.to(BF16).contiguous()


class LayerNorm(nn.LayerNorm):
    """fp32-statistics LayerNorm on bf16 activations.  `deep_spec_augment` (set by
    register_deep_spec_augment_hooks) is a callable -> (t0, t1, c0, c1) or None, evaluated per
    forward; the mask is applied inside the same kernel."""

    deep_spec_augment = None
    wft_fp32 = False  # Whisper.set_compute_dtype("fp32"): the fp32 compute mode (engine/ops32.py)

    def forward(self, x: Tensor) -> Tensor:
        mask = None
        if self.deep_spec_augment is not None and self.training and x.dim() == 3:
            m = self.deep_spec_augment()
            if m is not None:
                mask = (x.shape[1],) + tuple(m)
        if self.wft_fp32:
            return ops32.LayerNormFn.apply(x, self.weight, self.bias, self.eps, mask)
        return ops.LayerNormFn.apply(_to_bf16(x), self.weight, self.bias, self.eps, mask)

    def fork(self, x: Tensor):
        """(ln(x), residual alias of x) with the residual-gradient add fused in the backward."""
        if self.wft_fp32 or self._forward_hooks or self._forward_pre_hooks:
            return self(x), x  # fp32 mode: nothing fused; hooks: honour user hooks (the reference registers them on attn_ln)
        mask = None
        if self.deep_spec_augment is not None and self.training and x.dim() == 3:
            m = self.deep_spec_augment()
            if m is not None:
                mask = (x.shape[1],) + tuple(m)
        return ops.LayerNormForkFn.apply(_to_bf16(x), self.weight, self.bias, self.eps, mask)


class Linear(nn.Linear):
    """whisper.model.Linear: the LoRA target type (model/lora.py:46,55)."""

    def _group(self) -> ops.LinearGroup:
        g = self.__dict__.get("_wft_group")
        if g is None:
            g = ops.LinearGroup()
            self.__dict__["_wft_group"] = g
        return g

    def base_weight(self) -> Tensor:
        """The trainable/frozen base matrix (``parametrizations.weight.original`` under LoRA)."""
        if "parametrizations" in self._modules:
            return self.parametrizations.weight.original
        return self.weight

    def lora_spec(self) -> Optional[ops.LoraSpec]:
        if "parametrizations" not in self._modules:
            return None
        return self.parametrizations.weight[0].spec(self.training)

    wft_fp32 = False

    def forward(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        shape = x.shape
        if self.wft_fp32:
            y = ops32.linear(x, self.base_weight(), self.bias, self.lora_spec())
            return y if residual is None else ops32.add(y, residual.view(y.shape))
        y = ops.linear(_as2d(_to_bf16(x)), self._group(), [self.base_weight()], [self.bias], [self.lora_spec()],
                       residual=None if residual is None else _as2d(residual))
        return y.view(*shape[:-1], y.shape[-1])


class Conv1d(nn.Conv1d):
    """whisper.model.Conv1d — parameters only; evaluated by ops.ConvStemFn."""


class MultiHeadAttention(nn.Module):
    def __init__(self, n_state: int, n_head: int):
        super().__init__()
        assert n_state % n_head == 0 and n_state // n_head == 64, "libwft attention kernels are head_dim-64 (all Whisper sizes)"
        self.n_head = n_head
        self.query = Linear(n_state, n_state)
        self.key = Linear(n_state, n_state, bias=False)
        self.value = Linear(n_state, n_state)
        self.out = Linear(n_state, n_state)
        # self-attention: the softmax scale * log2(e) rides in the q rows of the fused projection's FORWARD shadow (ops.QK_PRESCALE)
        self._qkv_group = ops.LinearGroup(fwd_scales=(ops.QK_ALPHA, 1.0, 1.0) if ops.QK_PRESCALE else None)
        self._kv_group = ops.LinearGroup()
        self.wft_fp32 = False

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None, kv_cache: Optional[dict] = None,
                residual: Optional[Tensor] = None):
        """x [B,T,d] (already layer-normed); xa: encoder output for cross-attention; mask is not None
        => causal (the decoder's -inf upper-triangular buffer).  Returns (out, None) like upstream;
        `residual` (engine extension) is added inside the out-projection epilogue."""
        if kv_cache:
            raise NotImplementedError("kv_cache is an inference feature; the training/eval path is teacher-forced")
        B, T, d = x.shape
        if self.wft_fp32:  # fp32 mode: separate projections, strided per-head GEMMs, materialised probabilities
            src = x if xa is None else xa
            o = ops32.AttentionFn.apply(self.query(x), self.key(src), self.value(src), self.n_head, xa is None and mask is not None)
            return self.out(o, residual=residual), None
        x2 = _as2d(_to_bf16(x))
        if xa is None:
            lin = [self.query, self.key, self.value]
            qkv = ops.linear(x2, self._qkv_group, [m.base_weight() for m in lin], [m.bias for m in lin],
                             [m.lora_spec() for m in lin])
            o = ops.SelfAttnFn.apply(qkv.view(B, T, 3 * d), self.n_head, mask is not None, self._qkv_group.fwd_scales is not None)
        else:
            Ta = xa.shape[1]
            q = self.query(x2)
            lin = [self.key, self.value]
            # every decoder block projects the SAME encoder output: their input gradients are summed inside the backward-data
            # GEMMs (ops.GradAccum) instead of by 31 elementwise adds; the fork node and its accumulator ride on the tensor
            # itself, so any decoder class (plain, checkpointed, stochastic depth) shares them without knowing
            acc = None
            if _XA_ACCUM and torch.is_grad_enabled() and xa.requires_grad and xa.dtype == torch.bfloat16:
                xa, acc = ops.grad_fork(xa)
            kv = ops.linear(_as2d(_to_bf16(xa)), self._kv_group, [m.base_weight() for m in lin], [m.bias for m in lin],
                            [m.lora_spec() for m in lin], dx_accum=acc)
            o = ops.CrossAttnFn.apply(q.view(B, T, d), kv.view(B, Ta, 2 * d), self.n_head)
        out = self.out(o.view(B * T, d), residual=None if residual is None else _as2d(residual))
        return out.view(B, T, d), None


class MLP(nn.Sequential):
    """Sequential(Linear(d,4d), GELU(), Linear(4d,d)) — keys mlp.0 / mlp.2 — evaluated as two GEMMs
    with GELU fused in the first epilogue and gelu' fused in the backward-data epilogue."""

    wft_fp32 = False

    def forward(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        fc1, fc2 = self[0], self[2]
        shape = x.shape
        if self.wft_fp32:
            return fc2(ops32.GeluFn.apply(fc1(x)), residual=residual)
        x2 = _as2d(_to_bf16(x))
        pre, act, codes = ops.linear(x2, fc1._group(), [fc1.base_weight()], [fc1.bias], [fc1.lora_spec()], gelu_out=True,
                                     gelu_next_n=fc2.base_weight().shape[0])  # (not fc2.weight: on an adapted Linear that property MATERIALISES W + s B A)
        y = ops.linear(act, fc2._group(), [fc2.base_weight()], [fc2.bias], [fc2.lora_spec()],
                       residual=None if residual is None else _as2d(residual), gelu_pre=pre, gelu_codes=codes)
        return y.view(shape)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_state: int, n_head: int, cross_attention: bool = False):
        super().__init__()
        self.attn = MultiHeadAttention(n_state, n_head)
        self.attn_ln = LayerNorm(n_state)
        self.cross_attn = MultiHeadAttention(n_state, n_head) if cross_attention else None
        self.cross_attn_ln = LayerNorm(n_state) if cross_attention else None
        n_mlp = n_state * 4
        self.mlp = MLP(Linear(n_state, n_mlp), nn.GELU(), Linear(n_mlp, n_state))
        self.mlp_ln = LayerNorm(n_state)

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None, kv_cache: Optional[dict] = None):
        # x = x + attn(attn_ln(x)); x = x + cross_attn(cross_attn_ln(x), xa); x = x + mlp(mlp_ln(x))
        h, r = self.attn_ln.fork(x)
        x = self.attn(h, mask=mask, kv_cache=kv_cache, residual=r)[0]
        if self.cross_attn is not None:
            h, r = self.cross_attn_ln.fork(x)
            x = self.cross_attn(h, xa, kv_cache=kv_cache, residual=r)[0]
        h, r = self.mlp_ln.fork(x)
        return self.mlp(h, residual=r)


class AudioEncoder(nn.Module):
    def __init__(self, n_mels: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.conv1 = Conv1d(n_mels, n_state, kernel_size=3, padding=1)
        self.conv2 = Conv1d(n_state, n_state, kernel_size=3, stride=2, padding=1)
        self.register_buffer("positional_embedding", sinusoids(n_ctx, n_state))
        self.blocks: Iterable[ResidualAttentionBlock] = nn.ModuleList(
            [ResidualAttentionBlock(n_state, n_head) for _ in range(n_layer)]
        )
        self.ln_post = LayerNorm(n_state)
        self._stem_cache: dict = {}
        self.wft_fp32 = False

    def stem(self, x: Tensor) -> Tensor:
        """gelu(conv1) -> gelu(conv2) -> permute -> + positional_embedding; x f32 [B, n_mels, 2*n_ctx]."""
        if x.dim() != 3 or x.shape[1] != self.conv1.in_channels:
            raise ValueError(f"expected mel [B, {self.conv1.in_channels}, T], got {tuple(x.shape)}")
        assert x.shape[2] // 2 == self.positional_embedding.shape[0] and x.shape[2] % 2 == 0, "incorrect audio shape"
        if self.wft_fp32:
            return ops32.ConvStemFn.apply(x.float(), self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                          self.positional_embedding)
        c_pad = K.round_up(self.conv1.in_channels, 128)
        mel_t = K.mel_to_tmajor(x.float(), c_pad)
        return ops.ConvStemFn.apply(mel_t, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                    self.positional_embedding, self._stem_cache)

    def forward(self, x: Tensor):
        x = self.stem(x)
        for block in self.blocks:
            x = block(x)
        return self.ln_post(x)


class TextDecoder(nn.Module):
    def __init__(self, n_vocab: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, n_state)
        self.positional_embedding = nn.Parameter(torch.empty(n_ctx, n_state))
        self.blocks: Iterable[ResidualAttentionBlock] = nn.ModuleList(
            [ResidualAttentionBlock(n_state, n_head, cross_attention=True) for _ in range(n_layer)]
        )
        self.ln = LayerNorm(n_state)
        mask = torch.empty(n_ctx, n_ctx).fill_(-np.inf).triu_(1)
        self.register_buffer("mask", mask, persistent=False)
        self._logit_group = ops.LinearGroup()
        self.wft_fp32 = False

    def embed(self, tokens: Tensor) -> Tensor:
        if self.wft_fp32:
            return ops32.EmbedFn.apply(tokens, self.token_embedding.weight, self.positional_embedding)
        return ops.EmbedFn.apply(tokens, self.token_embedding.weight, self.positional_embedding)

    def hidden(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None) -> Tensor:
        x = self.embed(x)
        for block in self.blocks:
            x = block(x, xa, mask=self.mask, kv_cache=kv_cache)
        return self.ln(x)

    def padded_logits(self, h: Tensor) -> Tensor:
        """bf16 [B*S, V rounded up to 128] (columns >= V are zero-weight padding); fp32 mode: f32 [B*S, V]."""
        if self.wft_fp32:
            return ops32.TiedLogitsFn.apply(h, self.token_embedding.weight)
        return ops.TiedLogitsFn.apply(_as2d(h), self.token_embedding.weight, self._logit_group)

    def logits_from_hidden(self, h: Tensor) -> Tensor:
        B, S, _ = h.shape
        V = self.token_embedding.weight.shape[0]
        if self.wft_fp32:
            return self.padded_logits(h).view(B, S, V)
        return self.padded_logits(h)[:, :V].float().view(B, S, V)

    def forward(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None):
        return self.logits_from_hidden(self.hidden(x, xa, kv_cache))


def check_amp_request(model, mixed_precision: bool, mp_dtype: str) -> None:
    """The engine has two compute modes: bf16 (MFMA inputs bf16, fp32 accumulation, fp32 master weights — the arithmetic of the
    reference's `autocast(dtype=bfloat16)`, model/model_utils.py:37-48,64) and fp32 (engine/ops32.py — the reference with AMP
    off).  A request that does not match the model's mode, or fp16 autocast, raises instead of silently computing with
    different numerics; other modules (the reference's tests drive train_step with plain nn.Modules) are not concerned."""
    if not isinstance(model, Whisper):
        return
    if not mixed_precision:
        if model.compute_dtype != "fp32":
            raise ValueError("training.mixed_precision_training: False asks for fp32 compute but this model is in its bf16 mode: "
                             "call model.set_compute_dtype('fp32') first (scripts/finetune.py does) — the engine never computes "
                             "bf16 silently where the reference computes fp32")
        return
    if model.compute_dtype == "fp32":
        raise ValueError("the model is in its fp32 compute mode but mixed_precision_training: True asks for 16-bit autocast: "
                         "call model.set_compute_dtype('bf16')")
    if mp_dtype == "fp16":
        raise ValueError("training.mp_dtype: fp16 asks for fp16 autocast + loss scaling; the libwft engine computes in bf16 "
                         "(same 16-bit storage, 8-bit exponent, no GradScaler needed): set mp_dtype: bf16 "
                         "(scripts/finetune.py does this rewrite itself, with a warning)")


class Whisper(nn.Module):
    compute_dtype = "bf16"

    def set_compute_dtype(self, dtype: str) -> "Whisper":
        """"bf16" (default: the throughput path) or "fp32" (parity mode, engine/ops32.py).  Call again after swapping in
        new sub-modules (scripts/finetune.py does, after the checkpointed encoder / decoder classes and layer resizing)."""
        if dtype not in ("bf16", "fp32"):
            raise ValueError(f"compute dtype must be 'bf16' or 'fp32', got {dtype!r}")
        self.compute_dtype = dtype
        self.__dict__.pop("_wft_bias_params", None)
        for m in self.modules():
            if hasattr(m, "wft_fp32"):
                m.wft_fp32 = dtype == "fp32"
        return self

    def __init__(self, dims: ModelDimensions):
        super().__init__()
        self.dims = dims
        self.encoder = AudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer)
        self.decoder = TextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer)
        all_heads = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
        all_heads[dims.n_text_layer // 2:] = True
        self.register_buffer("alignment_heads", all_heads.to_sparse(), persistent=False)

    def train(self, mode: bool = True):
        """nn.Module.train for a tree whose modules all use the stock method (every module of this package does): one flat
        walk that sets the flag, instead of a recursive call and an nn.Module.__setattr__ per module.  train_step calls
        model.train() at the top of every step (model/model_utils.py:33 in the reference): 2 500 modules on large-v3,
        3-4 ms of host time while the GPU has nothing queued.  Any overriding sub-module sends the call down the stock path."""
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        stack, seen, flat = [self], set(), []
        while stack:
            m = stack.pop()
            if id(m) in seen:
                continue
            seen.add(id(m))
            if m is not self and type(m).train is not nn.Module.train:
                return super().train(mode)
            flat.append(m)
            stack.extend(c for c in m._modules.values() if c is not None)
        for m in flat:
            m.__dict__["training"] = mode
        return self

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def is_multilingual(self):
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self):
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def embed_audio(self, mel: Tensor):
        return self.encoder(mel)

    def logits(self, tokens: Tensor, audio_features: Tensor):
        return self.decoder(tokens, audio_features)

    def forward(self, mel: Tensor, tokens: Tensor, targets: Optional[Tensor] = None, label_smoothing: float = 0.0) -> Tensor:
        """logits f32 [B, S, V] — or, when `targets` is given (engine extension used by train_step, also
        through a DDP wrapper), the fused label-smoothed cross-entropy loss."""
        ops.reset_colsums()  # stale fused bias-gradient entries of an earlier backward pass (engine/ops.py)
        # a hint for the producer kernels (captured per autograd node at forward time): with every bias frozen (a LoRA run)
        # nobody will ask for the fused bias-gradient column sums
        # (the walk over named_parameters() costs 7 ms on large-v3 — at the start of a step, with the GPU idle: the bias list is
        # cached; set_compute_dtype(), which the entrypoint calls after every module swap, drops it.  A stale list can only
        # cost speed: a consumer that finds no fused column sums forms its bias gradient with wft_colsum_bf16)
        biases = self.__dict__.get("_wft_bias_params")
        if biases is None:
            biases = self.__dict__["_wft_bias_params"] = [p for n, p in self.named_parameters() if n.endswith("bias")]
        ops.BIAS_GRADS[0] = any(p.requires_grad for p in biases)
        if targets is not None:
            return self.forward_loss(mel, tokens, targets, label_smoothing)
        return self.decoder(tokens, self.encoder(mel))

    def forward_loss(self, mel: Tensor, tokens: Tensor, targets: Tensor, label_smoothing: float = 0.0) -> Tensor:
        """Fused equivalent of F.cross_entropy(model(mel, tokens).transpose(1, 2), targets, label_smoothing)
        (model/model_utils.py:65-66) that never materialises fp32 logits."""
        h = self.decoder.hidden(tokens, self.encoder(mel))
        if self.compute_dtype == "fp32":
            return ops32.CrossEntropyFn.apply(self.decoder.padded_logits(h), targets.reshape(-1).contiguous(), float(label_smoothing))
        return ops.FusedCEFn.apply(self.decoder.padded_logits(h), targets.reshape(-1), self.dims.n_vocab, float(label_smoothing))


def init_random_(model: Whisper, seed: int = 0, std: float = 0.02) -> Whisper:
    """SURVEY.md §8d random init: matrices/embeddings N(0, std^2), biases 0, LN gamma 1 / beta 0,
    encoder positions = sinusoids.  Same draw order as oracle.whisper_oracle.init_params."""
    from collections import OrderedDict

    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    new = OrderedDict()
    for name, t in sd.items():
        if name == "encoder.positional_embedding":
            new[name] = sinusoids(*t.shape)
        elif name.endswith("_ln.weight") or name in ("encoder.ln_post.weight", "decoder.ln.weight"):
            new[name] = torch.ones_like(t)
        elif name.endswith(".bias"):
            new[name] = torch.zeros_like(t)
        else:
            new[name] = None
    # matrices in the oracle's order
    order = ["encoder.conv1.weight", "encoder.conv2.weight"]
    d = model.dims

    def block_names(prefix, cross):
        out = []
        for a in ["attn"] + (["cross_attn"] if cross else []):
            out += [f"{prefix}.{a}.query.weight", f"{prefix}.{a}.key.weight", f"{prefix}.{a}.value.weight", f"{prefix}.{a}.out.weight"]
        out += [f"{prefix}.mlp.0.weight", f"{prefix}.mlp.2.weight"]
        return out

    for i in range(d.n_audio_layer):
        order += block_names(f"encoder.blocks.{i}", False)
    order += ["decoder.token_embedding.weight", "decoder.positional_embedding"]
    for i in range(d.n_text_layer):
        order += block_names(f"decoder.blocks.{i}", True)
    for name in order:
        new[name] = torch.randn(*sd[name].shape, generator=g) * std
    missing = [k for k, v in new.items() if v is None]
    assert not missing, missing
    model.load_state_dict(new)
    return model
