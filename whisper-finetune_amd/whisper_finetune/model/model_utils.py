"""Train step and model wrappers — same names and semantics as the reference's
`model/model_utils.py` (train_step :23-127, save_model :130-135, resize_whisper_layers :138-206,
infinite_iter :209-217, StochasticDepthMixin :220-250, CheckpointedStochastic{AudioEncoder,
TextDecoder} :253-327, register_deep_spec_augment_hooks :382-437), driving the libwft engine.

What is different, and why it is still a drop-in:
  * train_step uses `model.forward_loss(...)` (logits GEMM + label-smoothed CE fused, bf16 logits
    never up-cast / re-read 4x) when the un-wrapped model offers it; any other nn.Module takes the
    reference's `model(x, y_in)` + F.cross_entropy route unchanged;
  * autocast is entered for the device the model lives on (the reference hard-codes "cuda");
  * deep-SpecAugment draws its mask spans on the host in torchaudio's order and hands them to the
    LayerNorm kernel instead of running permute + 2 masked_fill + permute per block.
"""
from __future__ import annotations

import copy
import io
import os
from dataclasses import asdict
from functools import partial
from typing import Callable, Iterator, Optional, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils.checkpoint import checkpoint

import whisper_finetune.runtime as rt
from whisper_finetune.data import transforms as T  # the reference's `import torchaudio.transforms as T` (model_utils.py:10)
from whisper_finetune.engine.whisper_model import AudioEncoder, LayerNorm, TextDecoder, Whisper, check_amp_request

# PyTorch-ROCm words it "HIP error: an illegal memory access ..."; the CUDA spelling is the reference's (model_utils.py:76)
_ILLEGAL = ("HIP error: an illegal memory", "CUDA error: an illegal memory")


def _micro_batch_loss(model, x, y_in, y_out, label_smoothing: float) -> Tensor:
    if hasattr(rt.unwrap_model(model), "forward_loss"):
        # engine model (possibly inside a DDP wrapper): fused logits + CE, same value as the two-step form
        return model(x, y_in, targets=y_out, label_smoothing=label_smoothing)
    logits = model(x, y_in)
    return F.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing=label_smoothing)


def train_step(
    model,
    train_iter: Iterator,
    optimizer: torch.optim.Optimizer,
    lr_scheduler,
    t_config: dict,
    lora_tracker=None,
    step: int = None,
    scaler: Optional[torch.amp.GradScaler] = None,
) -> float:
    """One optimizer step = `accum_grad_steps` micro-batches.  Returns the sum over micro-batches of
    (mean CE / accum) on this rank (not all-reduced), exactly as the reference."""
    model.train()
    mixed = t_config["mixed_precision_training"]
    accum = t_config["accum_grad_steps"]
    max_grad_norm = t_config["max_grad_norm"]
    fp16 = mixed and t_config["mp_dtype"] == "fp16"
    amp_dtype = torch.float16 if t_config["mp_dtype"] == "fp16" else torch.bfloat16
    label_smoothing = t_config.get("label_smoothing", 0.0)
    is_lora_run = t_config.get("is_lora_run", False)
    if fp16 and scaler is None:
        raise ValueError("fp16 mixed precision training requires a persistent GradScaler.")
    check_amp_request(rt.unwrap_model(model), mixed, t_config["mp_dtype"])
    if not fp16:
        scaler = None

    device = next(model.parameters()).device
    graphed = _graphed_micro_batch(model, t_config, mixed, amp_dtype, label_smoothing, accum, scaler)
    total_loss = 0.0
    # The reference reads every micro-batch loss back with `loss.item()` (model/model_utils.py:73) — a host sync in front of clip +
    # optimizer.step(), whose Python (pointer tables over 1 259 tensors) then runs while the GPU idles: 4 ms per headline step
    # (bench.py hand_rolled_ms_per_step).  On a HIP device without a GradScaler the value is copied to a pinned slot behind an
    # event instead; the clip / optimizer / scheduler / zero_grad launches are enqueued, and only then does the host wait — for
    # the EVENT, i.e. for the backward pass, not for the optimizer kernels.  Same returned float, same order of additions.
    defer = device.type == "cuda" and scaler is None and t_config.get("wft_defer_loss_readback", True)
    pending = []
    for micro in range(accum):
        last = micro == accum - 1
        for attempt in range(3):
            try:
                x, y_in, y_out = next(train_iter)
                x = x.to(device, non_blocking=True)
                y_in = y_in.to(device, non_blocking=True)
                y_out = y_out.to(device, non_blocking=True)
                if graphed is not None:  # training.wft_hip_graph: forward + loss + backward as ONE graph launch (engine/graph.py)
                    if defer:
                        pending.append(_async_readback(graphed(x, y_in, y_out), device, micro, accum))
                    else:
                        total_loss += graphed(x, y_in, y_out).item()
                    break
                # the backward pass that runs beside the bucketed all-reduce gets per-tile launches: the mode is noted by the autograd
                # nodes while the forward runs on this thread and handed to their backward kernels per call (runtime.exchange_launch_mode)
                with rt.maybe_no_sync(model, enabled=rt.IS_DISTRIBUTED and not last), \
                        rt.exchange_launch_mode(rt.IS_DISTRIBUTED and last and device.type == "cuda"):
                    with torch.autocast(device_type=device.type, enabled=mixed, dtype=amp_dtype):
                        loss = _micro_batch_loss(model, x, y_in, y_out, label_smoothing) / accum
                    (scaler.scale(loss) if scaler else loss).backward()
                if defer:
                    pending.append(_async_readback(loss.detach(), device, micro, accum))
                else:
                    total_loss += loss.item()
                break
            except RuntimeError as err:
                # the reference retries sporadic illegal-memory-access errors 3x on one GPU and
                # aborts immediately under DDP (a retry would desynchronise the ranks)
                if not any(s in str(err) for s in _ILLEGAL):
                    raise
                if rt.IS_DISTRIBUTED:
                    print("Caught illegal memory access under DDP; aborting instead of retrying.")
                    raise
                print(f"Caught illegal memory access, retry {attempt + 1}")
                if attempt == 2:
                    print("Max retries reached. Something is wrong.")
                    raise

    if scaler:
        scaler.unscale_(optimizer)

    val_steps, train_steps = t_config.get("val_steps"), t_config.get("train_steps")
    if is_lora_run and step is not None and val_steps is not None and (step % val_steps == 0 or step == train_steps):
        from whisper_finetune.model.lora import log_lora_debug_info

        log_lora_debug_info(rt.unwrap_model(model), step=step, tracker=lora_tracker, log_to_wandb=rt.IS_MAIN)

    if scaler is None and hasattr(optimizer, "fuse_clip_grad_norm"):
        # libwft optimizers compute the global norm and apply the clip coefficient inside their update kernels
        optimizer.fuse_clip_grad_norm(max_grad_norm)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)

    if is_lora_run and lora_tracker is not None:
        lora_tracker.snapshot()

    if scaler:
        before = scaler.get_scale()
        scaler.step(optimizer)
        scaler.update()
        if scaler.get_scale() >= before:  # the step was not skipped for overflow
            lr_scheduler.step()
    else:
        optimizer.step()
        lr_scheduler.step()
    # (graph mode: the captured backward adds into persistent gradient buffers — zeroed in place, never dropped; this holds for
    # every later step of a model that has captured graphs, also one that runs eagerly)
    if graphed is not None:
        graphed.zero_grads(optimizer)
    else:
        optimizer.zero_grad(set_to_none=not _has_graphs(model))
    for slot, ev in pending:  # (in micro-batch order: the reference's order of additions)
        ev.synchronize()
        total_loss += slot.item()
    return total_loss


_READBACK = {}  # (device index, accum) -> pinned f32 [accum]


def _async_readback(loss: Tensor, device, micro: int, accum: int):
    """-> (pinned one-element view that will hold `loss`, event behind the copy).  The slots of one optimizer step are distinct; the
    next step reuses them after its predecessor has read them."""
    key = (device.index, accum)
    buf = _READBACK.get(key)
    if buf is None:
        buf = _READBACK[key] = torch.empty(accum, dtype=torch.float32, pin_memory=True)
    slot = buf[micro:micro + 1]
    slot.copy_(loss.reshape(1).float(), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return slot, ev


def _has_graphs(model) -> bool:
    from whisper_finetune.engine import graph as G

    return G.has_graphs(rt.unwrap_model(model))


def _graphed_micro_batch(model, t_config, mixed, amp_dtype, label_smoothing, accum, scaler):
    """The model's GraphedMicroBatch if `training.wft_hip_graph` asks for one and the configuration can be captured, else None
    (with one loud message saying why).  Cached on the model object; rebuilt if label smoothing or the accumulation factor change."""
    if not t_config.get("wft_hip_graph", False):
        return None
    from whisper_finetune.engine import graph as G

    m = rt.unwrap_model(model)
    reason = None
    if m is not model:
        reason = "the model is wrapped (DDP): its reducer hooks are Python"
    elif not mixed or amp_dtype != torch.bfloat16 or scaler is not None:
        reason = "only bf16 mixed precision is captured"
    else:
        reason = G.why_not(m)
    if reason is not None:
        if not m.__dict__.get("_wft_graph_refused"):
            m.__dict__["_wft_graph_refused"] = True
            print(f"WARNING: training.wft_hip_graph is set but this run stays on the eager path: {reason}.")
        return None
    key = (float(label_smoothing), int(accum))
    ent = G.graphed_for(m)  # (kept off the module: save_model deep-copies it, CUDAGraph objects do not copy)
    if ent is None or ent[0] != key:
        G.set_graphed(m, key, G.GraphedMicroBatch(m, label_smoothing, accum, amp_dtype))
        ent = G.graphed_for(m)
    return ent[1]


def save_model(model, save_path: str) -> None:
    """fp16 state dict + dims, loadable by `whisper.load_model` (model/model_utils.py:130-135)."""
    m = copy.deepcopy(rt.unwrap_model(model)).half()
    torch.save({"model_state_dict": m.state_dict(), "dims": asdict(m.dims)}, save_path)


def _resample_block_list(blocks: torch.nn.ModuleList, target_layers: int) -> torch.nn.ModuleList:
    """Keep / drop / duplicate blocks proportionally so that exactly target_layers remain: block i is
    repeated floor((i+1)*t/n) - floor(i*t/n) times (model/model_utils.py:138-168)."""
    if target_layers < 1:
        raise ValueError(f"target_layers must be >= 1, got {target_layers}")
    n = len(blocks)
    if n < 1:
        raise ValueError("Cannot resize an empty block list")
    if target_layers == n:
        return blocks
    out = []
    for i, blk in enumerate(blocks):
        reps = (i + 1) * target_layers // n - i * target_layers // n
        out.extend([blk] + [copy.deepcopy(blk) for _ in range(reps - 1)] if reps > 0 else [])
    if len(out) != target_layers:
        raise RuntimeError(f"Layer resizing produced {len(out)} blocks, expected {target_layers}.")
    return torch.nn.ModuleList(out)


def resize_whisper_layers(model, target_encoder_layers: Optional[int] = None, target_decoder_layers: Optional[int] = None) -> bool:
    changed = False
    if target_encoder_layers is not None and target_encoder_layers != len(model.encoder.blocks):
        old = len(model.encoder.blocks)
        model.encoder.blocks = _resample_block_list(model.encoder.blocks, target_encoder_layers)
        model.dims.n_audio_layer = target_encoder_layers
        print(f"Resized encoder layers: {old} -> {target_encoder_layers}")
        changed = True
    if target_decoder_layers is not None and target_decoder_layers != len(model.decoder.blocks):
        old = len(model.decoder.blocks)
        model.decoder.blocks = _resample_block_list(model.decoder.blocks, target_decoder_layers)
        model.dims.n_text_layer = target_decoder_layers
        heads = torch.zeros(target_decoder_layers, model.dims.n_text_head, dtype=torch.bool)
        heads[target_decoder_layers // 2:] = True
        model.register_buffer("alignment_heads", heads.to_sparse(), persistent=False)
        print(f"Resized decoder layers: {old} -> {target_decoder_layers}")
        changed = True
    return changed


def infinite_iter(data_loader) -> Iterator:
    """Cycle a DataLoader forever, calling sampler.set_epoch(0, 1, 2, ...) (DistributedSampler reshuffle)."""
    epoch = 0
    while True:
        sampler = getattr(data_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        yield from data_loader
        epoch += 1


class StochasticDepthMixin:
    """Stochastic depth (https://arxiv.org/abs/1603.09382) — model/model_utils.py:220-250.

    The reference wraps every kept block in `checkpoint` (recompute in backward) to fit 24-80 GB cards.  An MI355X holds
    288 GB, so by default the activations simply stay resident (`recompute = False`: identical numbers, one forward pass
    less per step); set `.recompute = True` (finetune.py: training.wft_recompute) to get the reference's memory profile."""

    recompute = False

    def stochastic_depth(self, x: Tensor, layer: Callable[[Tensor], Tensor], p: float) -> Tensor:
        # the skip decision is a HOST draw from the default CPU generator (RNG parity with the reference)
        if self.training and p > 0.0 and torch.rand(1).item() < p:
            return x
        out = checkpoint(layer, x, use_reentrant=False) if self.recompute else layer(x)
        if self.training and p > 0.0:
            keep = 1.0 - p
            if keep <= 0.0:
                return x
            if x.is_cuda and x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16:
                from whisper_finetune.engine.ops import SdRescaleFn

                return SdRescaleFn.apply(x, out, keep)  # one fused pass
            if x.is_cuda and x.dtype == torch.float32 and getattr(self, "wft_fp32", False):
                from whisper_finetune.engine.ops32 import sd_rescale

                return sd_rescale(x, out, keep)  # the fp32 compute mode
            return x + (out - x) / keep  # block(x) already contains the skip connection
        return out


class CheckpointedStochasticAudioEncoder(StochasticDepthMixin, AudioEncoder):
    def __init__(self, n_mels: int, n_ctx: int, n_state: int, n_head: int, n_layer: int, stochastic_depth_prob: float):
        super().__init__(n_mels, n_ctx, n_state, n_head, n_layer)
        self.stochastic_depth_prob = stochastic_depth_prob

    def forward(self, x: Tensor):
        x = self.stem(x)  # gelu(conv1) -> gelu(conv2) -> permute -> + positional_embedding
        for block in self.blocks:
            x = self.stochastic_depth(x, partial(block), self.stochastic_depth_prob)
        return self.ln_post(x)


class CheckpointedStochasticTextDecoder(StochasticDepthMixin, TextDecoder):
    def __init__(self, n_vocab: int, n_ctx: int, n_state: int, n_head: int, n_layer: int, stochastic_depth_prob: float):
        super().__init__(n_vocab, n_ctx, n_state, n_head, n_layer)
        self.stochastic_depth_prob = stochastic_depth_prob

    def hidden(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None) -> Tensor:
        x = self.embed(x)
        for block in self.blocks:
            x = self.stochastic_depth(x, partial(block, xa=xa, mask=self.mask, kv_cache=kv_cache), self.stochastic_depth_prob)
        return self.ln(x)


def load_model_and_set_heads(model: Whisper, name: str, device: Union[str, torch.device] = "cpu", download_root: Optional[str] = None,
                             in_memory: bool = False) -> Whisper:
    """Load a {"dims", "model_state_dict"} checkpoint file into `model` and move it to `device`
    (model/model_utils.py:330-379).  The reference also accepts the official model names and downloads them with
    `whisper._download`; there is no network here, so a name that is not a file raises the reference's RuntimeError."""
    if not os.path.isfile(name):
        from whisper_finetune.engine.whisper_model import MODEL_DIMS

        raise RuntimeError(f"Model {name} not found; available models = {sorted(MODEL_DIMS)} (official checkpoints are "
                           "downloaded by openai-whisper, which this build does not ship: pass a checkpoint path)")
    if in_memory:
        with open(name, "rb") as fh:
            blob = fh.read()
        checkpoint = torch.load(io.BytesIO(blob), map_location=device)
    else:
        checkpoint = torch.load(name, map_location=device)
    model.load_state_dict(checkpoint["model_state_dict"])
    return model.to(device)


def register_deep_spec_augment_hooks(model, time_mask_param: int, freq_mask_param: int, p: float = 1.0,
                                     layer_indices: Optional[list] = None) -> None:
    """SpecAugment on the normalised features after `attn_ln` of encoder blocks (all but the last by
    default): one time span <= time_mask_param and one channel span <= freq_mask_param are zeroed for
    the whole batch; on/off is decided once per encoder forward with probability p (model/model_utils.py:382-437).

    The maskers are built from the module-level namespace `T` like the reference's.  On the engine's LayerNorm with the
    native maskers the two spans are DRAWN here (same generator, same order) and applied inside the LayerNorm kernel;
    any other `attn_ln` module or masker type gets the reference's forward hook (permute -> time -> freq -> permute)."""
    p = float(p)
    if not 0.0 <= p <= 1.0:
        raise ValueError(f"deep_spec_augment p must be between 0 and 1, got {p}")
    time_mask = T.TimeMasking(time_mask_param=time_mask_param)
    freq_mask = T.FrequencyMasking(freq_mask_param=freq_mask_param)
    native = isinstance(time_mask, T._AxisMasking) and isinstance(freq_mask, T._AxisMasking)
    n_blocks = len(model.encoder.blocks)
    state = {"apply": False}

    def decide(module, inputs):
        # kept until the next encoder forward so a checkpoint recompute sees the same on/off state
        state["apply"] = True if p >= 1.0 else False if p <= 0.0 else torch.rand(1).item() < p

    def draw():
        if not state["apply"]:
            return None
        n_ctx, n_state = model.encoder.positional_embedding.shape
        t0, t1 = time_mask.draw(n_ctx)    # time mask first ...
        c0, c1 = freq_mask.draw(n_state)  # ... then the "frequency" (= channel) mask
        return t0, t1, c0, c1

    def norm_hook(module, inputs, output):
        if module.training and state["apply"]:
            x = output.permute(0, 2, 1)  # [B, d, T]: "frequency" = channel axis, time last
            x = freq_mask(time_mask(x))
            return x.permute(0, 2, 1)
        return output

    if layer_indices is None:
        layer_indices = range(n_blocks - 1)
    for idx in layer_indices:
        if idx >= n_blocks:
            raise ValueError(f"Layer index {idx} out of range")
        if idx == n_blocks - 1:
            continue  # never augment the last block: let the model recover
        ln = model.encoder.blocks[idx].attn_ln
        if native and isinstance(ln, LayerNorm):
            ln.deep_spec_augment = draw
        else:
            ln.register_forward_hook(norm_hook)
    model.encoder.register_forward_pre_hook(decide)
