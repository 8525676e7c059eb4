"""torch.autograd.Function wrappers: every forward/backward below is a sequence of libwft
kernel launches (kernels.py) — PyTorch only provides memory, streams and the autograd graph.

Layout conventions: activations are bf16, contiguous, 2-D [rows, features] (rows = B*T);
parameters stay fp32 (master weights, as the reference's AMP keeps them — SURVEY.md §7
"Numerics contract") and are shadowed in bf16 by `LinearGroup`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import itertools
import threading
import weakref

import os
import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from .. import runtime as _rt
from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16
F32 = torch.float32

# Softmax scale folded into the q projection: the FORWARD bf16 shadow of the self-attention q weight and the q
# slice of the fused bias carry scale * log2(e) (LinearGroup.fwd_scales -> wft_cast_pad_transpose_f32_bf16 / wft_lora_merge /
# wft_lora_refresh_mt `fwd_scale`, applied in fp32 before the one bf16 rounding), so the q third of the fused QKV output IS the exp2
# exponent's left operand and no attention kernel multiplies its scores (wft_attn_args.q_prescaled; -4 % on the three encoder kernels,
# profiles/r06_attn_prescale.md).  The backward-data shadow W^T stays unscaled and the dQ kernel returns the gradient w.r.t. the unscaled
# projection output, so dx, dW, db and the LoRA gradients are computed exactly as before.  Reference: whisper's `q * scale`
# (MultiHeadAttention.qkv_attention, reached from model/model_utils.py:283-285,320-322).  WFT_QK_PRESCALE=0: the old path (A/B runs).
QK_PRESCALE = os.environ.get("WFT_QK_PRESCALE", "1") != "0"
QK_ALPHA = float(torch.tensor(64 ** -0.5, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32))

# bumped by optimizers that update parameters through raw pointers (torch's in-place ops bump
# tensor._version themselves); part of every shadow-cache key.
_SHADOW_EPOCH = [0]


def bump_shadow_epoch():
    _SHADOW_EPOCH[0] += 1


# torch's fused / foreach optimizers update parameters WITHOUT bumping `_version` (observed with
# AdamW(fused=True): p._version stays 0 across steps), so the version part of the key cannot see them.
# A torch.optim.Optimizer.step() therefore invalidates the shadows through a global post-hook — but only a step of an
# optimizer that owns at least one parameter the engine has shadowed (an EMA / teacher optimizer over other tensors used to
# force a rebuild of every shadow per step of its own).
_SHADOWED = {}          # id(tensor) -> weakref: fp32 parameters (and conv weights) that have bf16 shadows somewhere in the process
_SHADOWED_GEN = [0]     # bumped whenever the registry changes (optimizers re-classify themselves lazily)


def note_shadowed(tensors) -> None:
    for t in tensors:
        if t is None or not t.requires_grad:
            continue
        r = _SHADOWED.get(id(t))
        if r is None or r() is not t:
            key = id(t)

            def _gone(_, key=key):
                _SHADOWED.pop(key, None)
                _SHADOWED_GEN[0] += 1

            _SHADOWED[key] = weakref.ref(t, _gone)
            _SHADOWED_GEN[0] += 1


def _optimizer_post_hook(optimizer, args, kwargs):
    # (re-)classify when the set of shadowed parameters OR the optimizer's own parameter list has changed (add_param_group with
    # parameters that were already shadowed leaves the generation alone: the signature below catches it)
    sig = (_SHADOWED_GEN[0], len(optimizer.param_groups), sum(len(g["params"]) for g in optimizer.param_groups))
    owns = optimizer.__dict__.get("_wft_owns_shadowed")
    if owns is None or owns[0] != sig:
        hit = any((r := _SHADOWED.get(id(p))) is not None and r() is p for g in optimizer.param_groups for p in g["params"])
        owns = optimizer.__dict__["_wft_owns_shadowed"] = (sig, hit)
    if owns[1]:
        bump_shadow_epoch()


register_optimizer_step_post_hook(_optimizer_post_hook)


def _ver(t: Optional[torch.Tensor]):
    return None if t is None else (t.data_ptr(), t._version)


class LoraSpec:
    """Low-rank adapter of one Linear (minLoRA semantics):
    W_eff = W + scaling * B @ (A * mask);  A [r, in], B [out, r], mask [1, in] (already drawn)."""

    __slots__ = ("A", "B", "scaling", "mask", "draw_id", "owner", "__weakref__")
    _draws = itertools.count(1)

    def __init__(self, A, B, scaling: float, mask: Optional[torch.Tensor], draw_id: Optional[int] = None, owner=None):
        self.A, self.B, self.scaling, self.mask = A, B, float(scaling), mask
        # a freshly drawn dropout mask is a new tensor with _version 0 whose storage the caching allocator very likely
        # recycles from the previous micro-batch's mask: (data_ptr, _version) cannot tell two draws apart.  Every drawn
        # mask therefore carries a process-wide serial number that is part of the shadow-cache keys: its own, or — for masks
        # that are slices of a model's mask pool — the serial of the pool's draw (one per forward of the model; a checkpoint
        # recompute inside that forward sees the same masks under the same serial and reuses the shadows).
        self.draw_id = 0 if mask is None else (next(LoraSpec._draws) if draw_id is None else draw_id)
        self.owner = owner  # the adapter module (model/lora.py), for LoraRefreshPlan

    def key(self):
        return (_ver(self.A), _ver(self.B), self.draw_id, self.scaling)


class LinearGroup:
    """bf16 shadows for one or several Linear layers that share their input and are evaluated
    as ONE GEMM (q/k/v -> N = 3*d).  Rebuilt only when a parameter changed."""

    def __init__(self, fwd_scales: Optional[Sequence[float]] = None):
        self.key = self.bkey = None
        self.bias_aliased = False
        self.W = self.WT = self.bias = None
        self.lkey = None
        self.Am = self.AmT = self.Bb = self.BbT = None
        # per weight of the group: factor folded into the FORWARD shadow W (and the matching slice of the stacked bias) in fp32
        # before the bf16 rounding; the transposed shadow W^T (backward-data) and the rank-r LoRA operands stay unscaled.
        # Self-attention q/k/v groups: (QK_ALPHA, 1, 1) — see QK_PRESCALE above.  A constant of the group: not part of the keys.
        self.fwd_scales = None if fwd_scales is None or all(float(f) == 1.0 for f in fwd_scales) else tuple(float(f) for f in fwd_scales)

    def fwd_scale(self, i: int) -> float:
        return 1.0 if self.fwd_scales is None else self.fwd_scales[i]

    @staticmethod
    def _fbits(f: float) -> int:
        """f32 bits of a forward scale for the high half of a table field (0 = 1.0; wft.h wft_lora_refresh_mt / wft_mt_copy_f32)."""
        import struct

        return 0 if f == 1.0 else struct.unpack("<I", struct.pack("<f", f))[0] << 32

    @staticmethod
    def dims(weights: Sequence[torch.Tensor]):
        """(n, k, n_pad): out-features summed over the group, in-features, n rounded up to 128."""
        n = sum(w.shape[0] for w in weights)
        k = weights[0][0].numel()
        if k % 128 != 0:
            raise L.WftError(f"in_features={k} must be a multiple of 128 (pad the operand)")
        return n, k, K.round_up(n, 128)

    def shadows(self, weights, biases, need_t: bool, loras=None):
        """bf16 [Npad, K] (+ transposed) image of the stacked weights.  With `loras`, every adapted weight is written as
        W + s*B@(A*mask) by wft_lora_merge — the same effective weight minLoRA's parametrization hands to F.linear — so
        the forward and backward-data GEMMs of a LoRA run are exactly the plain ones."""
        want_t = need_t or self.WT is not None
        # torch's fused optimizers do not bump _version: trainable tensors are additionally keyed on the optimizer epoch;
        # FROZEN ones (LoRA base weights) are not, so their shadows survive optimizer steps
        live = any(t is not None and t.requires_grad for t in list(weights) + list(biases))
        lkey = None
        if loras is not None and any(sp is not None for sp in loras):
            lkey = tuple(None if sp is None else sp.key() for sp in loras)
            live = True
        key = (tuple(_ver(w) for w in weights), want_t, _SHADOW_EPOCH[0] if live else -1, lkey)
        # biases have a key of their own: frozen ones (a LoRA run) are copied once, not with every new dropout mask
        bkey = self._bias_key(biases)
        if (key != self.key or bkey != self.bkey) and lkey is None and live and _SHADOW_BATCH and _PLAIN_PLAN.epoch != _SHADOW_EPOCH[0] \
                and id(self) in _PLAIN_PLAN.groups and self.WT is not None:
            # every trainable group's shadows AND stacked bias vectors in one launch each; sets self.key / self.bkey if this
            # group took part
            _PLAIN_PLAN.refresh()
        if bkey != self.bkey:
            n, k, npad = self.dims(weights)
            if not any(b is not None for b in biases):
                self.bias = None
            elif len(weights) == 1 and n == npad and biases[0].dtype == F32 and biases[0].is_contiguous() and biases[0].data_ptr() % 16 == 0 \
                    and self.fwd_scales is None:
                self.bias = biases[0].detach()  # the parameter itself: nothing to copy, never stale
            else:
                if self.bias is None or self.bias.shape[0] != npad or self.bias_aliased:
                    self.bias = torch.zeros(npad, dtype=F32, device=weights[0].device)
                off = 0
                for i, (w, b) in enumerate(zip(weights, biases)):
                    if b is not None:
                        if self.fwd_scale(i) == 1.0:
                            self.bias[off:off + w.shape[0]].copy_(b.detach())
                        else:  # the forward GEMM adds the bias to the SCALED product
                            torch.mul(b.detach(), self.fwd_scale(i), out=self.bias[off:off + w.shape[0]])
                    off += w.shape[0]
            self.bias_aliased = self.bias is not None and any(b is not None and b.data_ptr() == self.bias.data_ptr() for b in biases)
            self.bkey = bkey
        if key != self.key:
            note_shadowed(list(weights) + list(biases) + ([x for sp in loras if sp is not None for x in (sp.A, sp.B)] if loras else []))
            n, k, npad = self.dims(weights)
            dev = weights[0].device
            if self.W is None or self.W.shape != (npad, k):
                self.W = torch.zeros((npad, k), dtype=BF16, device=dev)
                self.WT = None
            if want_t and self.WT is None:
                self.WT = torch.zeros((k, npad), dtype=BF16, device=dev)
            off = 0
            for i, w in enumerate(weights):
                o = w.shape[0]
                sp = loras[i] if lkey is not None else None
                if sp is None:
                    K.weight_shadow(w.detach(), o, k, want_t, out=self.W[off:off + o],
                                    out_t=self.WT[:, off:off + o] if want_t else None, fwd_scale=self.fwd_scale(i))
                else:
                    K.lora_merge(w.detach(), sp.B.detach(), sp.A.detach(), None if sp.mask is None else sp.mask.detach(), sp.scaling,
                                 rows_pad=o, cols_pad=k, out=self.W[off:off + o], out_t=self.WT[:, off:off + o] if want_t else None,
                                 fwd_scale=self.fwd_scale(i))
                off += o
            self.key = key
            if lkey is not None and None not in lkey:
                _note_lora_group(self, weights, loras)
            elif lkey is None and _SHADOW_BATCH and want_t and all(w.requires_grad for w in weights):
                _PLAIN_PLAN.note(self, weights, biases)
        return self.W, self.WT, self.bias

    @staticmethod
    def _bias_key(biases):
        return (tuple(_ver(b) for b in biases), _SHADOW_EPOCH[0] if any(b is not None and b.requires_grad for b in biases) else -1)

    def lora_shadows(self, weights, loras: Sequence[Optional[LoraSpec]]):
        """scaling*(A*mask) stacked [Rpad, K] and block-diagonal scaling*B [Npad, Rpad], both (+T) in bf16: the operands of the
        rank-r gradient GEMMs (backward only); one wft_lora_pack launch per adapter."""
        key = tuple(None if s is None else s.key() for s in loras) + (_SHADOW_EPOCH[0],)
        if key != self.lkey:
            n, k, npad = self.dims(weights)
            dev = weights[0].device
            rtot = sum(s.A.shape[0] for s in loras if s is not None)
            rpad = K.round_up(rtot, 128)
            if self.Am is None or self.Am.shape != (rpad, k) or self.Bb.shape != (npad, rpad):
                # rows >= rtot of Am and the off-diagonal blocks of Bb stay zero
                self.Am = torch.zeros((rpad, k), dtype=BF16, device=dev)
                self.AmT = torch.zeros((k, rpad), dtype=BF16, device=dev)
                self.Bb = torch.zeros((npad, rpad), dtype=BF16, device=dev)
                self.BbT = torch.zeros((rpad, npad), dtype=BF16, device=dev)
            ro = no = 0
            for w, s in zip(weights, loras):
                if s is not None:
                    K.lora_pack(s.A.detach(), None if s.mask is None else s.mask.detach(), s.B.detach(), s.scaling,
                                self.Am, self.AmT, self.Bb, self.BbT, ro, no)
                    ro += s.A.shape[0]
                no += w.shape[0]
            self.lkey = key
        return self.Am, self.AmT, self.Bb, self.BbT


class LoraRefreshPlan:
    """Every adapted Linear group of one model, refreshed by ONE wft_lora_refresh_mt launch per training forward (merged bf16
    shadows W + s B (A*mask), their transposes, and the rank-r gradient-GEMM operands) instead of wft_lora_merge +
    wft_lora_pack per Linear.  Owned by the model's mask pool (model/lora.py), which calls `refresh` right after it has drawn
    the masks.  Groups enter the plan the first time they go through the per-Linear path (`LinearGroup.shadows` notes them);
    after a refresh each group's cache keys are set to exactly what `shadows` / `lora_shadows` will compute for this forward,
    so those calls find their buffers current — and fall back to the per-Linear kernels on any difference (a parameter
    modified in between, a group that is not in the table yet, an adapter switched off)."""

    def __init__(self):
        self.groups = {}   # id(group) -> (group, weights, adapters)
        self.dirty = True
        self.items = []    # [(group, weights, adapters, packed)] in table order
        self.ptrs = None
        self.table = self.tile_start = None
        self.total_tiles = 0
        self.in_table = {}  # id(group) -> packed
        self.unfit = set()  # groups whose shapes / alignment the batched kernel does not take: per-Linear path for good

    def note(self, group: "LinearGroup", weights, adapters) -> None:
        ent = self.groups.get(id(group))
        if ent is None or len(ent[1]) != len(weights) or any(a is not b for a, b in zip(ent[1], weights)) \
                or any(a is not b for a, b in zip(ent[2], adapters)):
            self.groups[id(group)] = (group, tuple(weights), tuple(adapters))
            self.dirty = True
        elif not self.dirty and id(group) not in self.unfit and self.in_table.get(id(group)) != (group.Am is not None):
            self.dirty = True  # noted before its buffers existed (or before its pack operands did)

    def _build(self, mask_of) -> None:
        import struct

        rows, tiles, items, ptrs = [], [0], [], []
        dev = None
        for group, weights, adapters in self.groups.values():
            if group.W is None or group.WT is None or not all(a.enabled for a in adapters):
                continue
            n, k, npad = group.dims(weights)
            if k % 64 or any(w.shape[0] % 64 or w.data_ptr() % 16 or not w.is_contiguous() for w in weights) \
                    or any(a.lora_A.shape[0] > 64 for a in adapters):
                self.unfit.add(id(group))
                continue
            dev = weights[0].device
            packed = group.Am is not None
            rpad = group.Am.shape[0] if packed else 0
            off = ro = 0
            for wi, (w, ad) in enumerate(zip(weights, adapters)):
                o, r = w.shape[0], ad.lora_A.shape[0]
                m = mask_of(ad)
                rows.append([w.data_ptr(), o, k, ad.lora_B.data_ptr(), ad.lora_A.data_ptr(), 0 if m is None else m.data_ptr(), r,
                             struct.unpack("<I", struct.pack("<f", float(ad.scaling)))[0] | group._fbits(group.fwd_scale(wi)),
                             group.W.data_ptr() + 2 * off * group.W.stride(0), group.WT.data_ptr() + 2 * off,
                             group.W.stride(0), group.WT.stride(0)]
                            + ([group.Am.data_ptr(), group.AmT.data_ptr(), group.Bb.data_ptr(), group.BbT.data_ptr()] if packed
                               else [0, 0, 0, 0]) + [rpad, npad, ro, off])
                tiles.append(tiles[-1] + (o // 64) * (k // 64))
                ptrs.extend((w.data_ptr(), ad.lora_A.data_ptr(), ad.lora_B.data_ptr()))
                off += o
                ro += r
            items.append((group, weights, adapters, packed))
        self.items, self.ptrs, self.total_tiles = items, ptrs, tiles[-1]
        self.in_table = {id(it[0]): it[3] for it in items}
        if rows:
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.tile_start = torch.tensor(tiles, dtype=torch.int32).to(dev)
        else:
            self.table = self.tile_start = None
        self.dirty = False

    def refresh(self, mask_of, serial: int) -> None:
        """mask_of(adapter) -> this forward's mask (a slice of the pool's PERSISTENT buffer) or None; serial: the draw id the
        adapters' LoraSpecs carry for it."""
        if not self.groups:
            return
        for attempt in range(2):
            if self.dirty:
                self._build(mask_of)
            if self.table is None:
                return
            vers, cur = [], []
            for group, weights, adapters, packed in self.items:
                wv = tuple(_ver(w) for w in weights)
                av = tuple((_ver(a.lora_A), _ver(a.lora_B)) for a in adapters)
                vers.append((wv, av))
                for x, (ya, yb) in zip(wv, av):
                    cur.extend((x[0], ya[0], yb[0]))
            # (pk != (g.Am is not None): a group that entered the table before its pack operands existed must be re-tabled as
            # packed, or every backward falls back to one wft_lora_pack launch per adapter for good — ADVICE r2)
            stale = cur != self.ptrs or any(g.W is None or g.WT is None or pk != (g.Am is not None) or not all(a.enabled for a in ads)
                                            for g, _, ads, pk in self.items)
            if not stale:
                break
            self.dirty = True  # a parameter or a shadow buffer moved: rebuild the table once
        else:
            return
        K.lora_refresh_mt(self.table, self.tile_start, self.table.shape[0], self.total_tiles)
        epoch = _SHADOW_EPOCH[0]
        for (group, weights, adapters, packed), (wv, av) in zip(self.items, vers):
            lkey = tuple((a_v, b_v, 0 if mask_of(ad) is None else serial, float(ad.scaling)) for (a_v, b_v), ad in zip(av, adapters))
            group.key = (wv, True, epoch, lkey)
            if packed:
                group.lkey = lkey + (epoch,)


class PlainRefreshPlan:
    """The same single launch for the bf16 shadows (+ transposes) of every TRAINABLE, un-adapted Linear group of the process: after
    an optimizer step the first group that finds its shadow stale refreshes all of them (wft_lora_refresh_mt rows with rank 0:
    cast + transpose), instead of one wft_cast_pad_transpose launch per weight spread over the forward (513 for large-v3, 97 for
    base).  Groups and parameters are held weakly; a group whose weights moved, or whose shapes the kernel does not take, stays on
    the per-weight path."""

    def __init__(self):
        self.groups = {}  # id(group) -> (weakref(group), tuple(weakref(weight)))
        self.dirty = True
        self.epoch = None
        self.tables = []  # [(table, tile_start, total_tiles, items, device)]; items: [(weakref(group), weights)]
        self.ptrs = None
        self.in_table = set()

    def note(self, group: "LinearGroup", weights, biases=()) -> None:
        ent = self.groups.get(id(group))
        if ent is None or ent[0]() is not group or len(ent[1]) != len(weights) or any(r() is not w for r, w in zip(ent[1], weights)) \
                or len(ent[2]) != len(biases) or any((r is None) != (b is None) or (r is not None and r() is not b) for r, b in zip(ent[2], biases)):
            n, k, npad = group.dims(weights)
            if k % 64 or any(w.shape[0] % 64 or w.data_ptr() % 16 or not w.is_contiguous() or w.dtype != F32 for w in weights):
                return
            self.groups[id(group)] = (weakref.ref(group), tuple(weakref.ref(w) for w in weights),
                                      tuple(None if b is None else weakref.ref(b) for b in biases))
            self.dirty = True
        elif id(group) not in self.in_table:
            self.dirty = True  # noted before its transposed shadow existed

    def _live(self):
        out = []
        for key, (gref, wrefs, brefs) in list(self.groups.items()):
            g, ws, bs = gref(), [r() for r in wrefs], [None if r is None else r() for r in brefs]
            gone = g is None or any(w is None for w in ws) or not all(w.requires_grad for w in ws)  # dropped, or frozen since
            gone = gone or any(r is not None and b is None for r, b in zip(brefs, bs))
            if gone or g.W is None or g.WT is None:
                if gone:
                    del self.groups[key]
                continue
            out.append((g, ws, bs))
        return out

    @staticmethod
    def _stacks_biases(g, ws, bs) -> bool:
        """The group owns a stacked bias vector that has to follow its parameters (not the single-Linear case, where the
        parameter itself is the bias operand), and every bias is a contiguous f32 tensor."""
        return (g.bias is not None and not g.bias_aliased and len(bs) == len(ws) and any(b is not None for b in bs)
                and any(b is not None and b.requires_grad for b in bs)
                and all(b is None or (b.dtype == F32 and b.is_contiguous() and b.numel() == w.shape[0]) for w, b in zip(ws, bs))
                and g.bias.shape[0] >= sum(w.shape[0] for w in ws))

    def _build(self) -> None:
        by_dev = {}
        for g, ws, bs in self._live():
            by_dev.setdefault(ws[0].device, []).append((g, ws, bs))
        self.tables, self.ptrs = [], []
        self.in_table = {id(g) for items in by_dev.values() for g, _, _ in items}
        for dev, items in by_dev.items():
            rows, tiles, brows = [], [0], []
            for g, ws, bs in items:
                n, k, npad = g.dims(ws)
                off = 0
                if self._stacks_biases(g, ws, bs):
                    for wi, (w, b) in enumerate(zip(ws, bs)):
                        if b is not None:
                            brows.append([b.data_ptr(), g.bias.data_ptr() + 4 * off, w.shape[0] | g._fbits(g.fwd_scale(wi))])
                            self.ptrs.extend((b.data_ptr(), g.bias.data_ptr()))
                        off += w.shape[0]
                    off = 0
                for wi, w in enumerate(ws):
                    o = w.shape[0]
                    rows.append([w.data_ptr(), o, k, 0, 0, 0, 0, g._fbits(g.fwd_scale(wi)), g.W.data_ptr() + 2 * off * g.W.stride(0), g.WT.data_ptr() + 2 * off,
                                 g.W.stride(0), g.WT.stride(0), 0, 0, 0, 0, 0, npad, 0, off])
                    tiles.append(tiles[-1] + (o // 64) * (k // 64))
                    self.ptrs.append(w.data_ptr())
                    off += o
            self.tables.append((torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(tiles, dtype=torch.int32).to(dev),
                                tiles[-1], [(weakref.ref(g), ws, bs) for g, ws, bs in items], dev,
                                torch.tensor(brows, dtype=torch.int64).to(dev) if brows else None))
        self.dirty = False

    def refresh(self) -> None:
        epoch = _SHADOW_EPOCH[0]
        self.epoch = epoch
        for attempt in range(2):
            if self.dirty:
                self._build()
            cur, ok = [], True
            for table, tile_start, total, items, dev, btable in self.tables:
                for gref, ws, bs in items:
                    g = gref()
                    ok = ok and g is not None and g.W is not None and g.WT is not None and all(w.requires_grad for w in ws)
                    if ok and self._stacks_biases(g, ws, bs):  # (the same order _build appends in)
                        for b in bs:
                            if b is not None:
                                cur.extend((b.data_ptr(), g.bias.data_ptr()))
                    cur.extend(w.data_ptr() for w in ws)
            if ok and cur == self.ptrs:
                break
            self.dirty = True
        else:
            return
        for table, tile_start, total, items, dev, btable in self.tables:
            with torch.cuda.device(dev):
                K.lora_refresh_mt(table, tile_start, table.shape[0], total)
                if btable is not None:
                    K.mt_copy_f32(btable)
            for gref, ws, bs in items:
                g = gref()
                # only groups whose every weight is trainable were noted: `live` in shadows() is True for them
                g.key = (tuple(_ver(w) for w in ws), True, epoch, None)
                if btable is not None and self._stacks_biases(g, ws, bs):
                    g.bkey = g._bias_key(bs)


_PLAIN_PLAN = PlainRefreshPlan()
# WFT_SHADOW_BATCH=0: one cast/transpose launch per weight (A/B runs)
_SHADOW_BATCH = os.environ.get("WFT_SHADOW_BATCH", "1") != "0"


def _note_lora_group(group: "LinearGroup", weights, loras) -> None:
    owners = [getattr(sp, "owner", None) for sp in loras]
    if any(o is None for o in owners):
        return
    pool = getattr(owners[0], "_pool", None)
    if pool is None or any(getattr(o, "_pool", None) is not pool for o in owners) or getattr(pool, "plan", None) is None:
        return
    pool.plan.note(group, weights, owners)


class GradAccum:
    """Sums the input gradients of SEVERAL Linear consumers of one tensor inside their backward-data GEMMs.

    The encoder output feeds the key / value projections of every decoder block (whisper's
    MultiHeadAttention.forward with `xa`, reached from the reference's model_utils.py:320-322): autograd would add the
    32 gradients of large-v3 pairwise, 31 elementwise kernels over a [B*1500, d] tensor per step (3.6 ms at 68 clips).
    Here the consumers read the tensor through ONE fork node (`grad_fork`): every consumer's GEMM adds its product to a
    running sum in its epilogue (C = acc + residual, in place) and returns None; the fork's backward — which autograd
    runs only after every consumer of the fork that takes part in THIS backward pass has run — hands the sum (plus
    whatever other users of the forked tensor sent through autograd) to the producer of the tensor.  Nothing is counted,
    so partial backward passes (stochastic-depth skips, `autograd.grad(inputs=subset)`, two decoder passes over one
    encoder output, a consumer used twice) are summed exactly like autograd would; a sum left behind by a pass that
    never reached the fork (an exception, the fork not in `inputs`) is recognised by its graph-task id and dropped."""

    __slots__ = ("buf", "task")

    def __init__(self):
        self.buf, self.task = None, -1

    def _fresh(self):
        task = torch._C._current_graph_task_id()
        if task != self.task:  # a new backward pass: whatever an unfinished one left is stale
            self.buf, self.task = None, task

    def arrive(self, dx_fn):
        """dx_fn(residual) -> the consumer's dx, added to `residual` in place when one is given.  Returns None: the
        fork hands the sum over."""
        self._fresh()
        self.buf = dx_fn(self.buf)
        return None

    def take(self):
        self._fresh()
        out, self.buf = self.buf, None
        return out


class _GradForkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, accum):
        ctx.set_materialize_grads(False)
        ctx.accum, ctx.shape = accum, x.shape
        return x.detach()  # (same storage; not a view of x, so the fork cached on x is not a reference cycle)

    @staticmethod
    def backward(ctx, g):
        s = ctx.accum.take()
        if s is None:
            return g, None
        s = s.view(ctx.shape)
        return (s if g is None else s + g.to(s.dtype)), None


def grad_fork(x: torch.Tensor):
    """(x_fork, accum): the alias of `x` all accumulating consumers read, created by the first of them and cached on the
    tensor object (pass `x` on by identity: a different tensor object for the same data gets a fork of its own — correct,
    one more elementwise add in autograd)."""
    ent = getattr(x, "_wft_fork", None)
    if ent is None:
        acc = GradAccum()
        ent = x._wft_fork = (_GradForkFn.apply(x, acc), acc)
    return ent


# Gradient homes.  torch DDP with gradient_as_bucket_view=True (the reference's wrap, scripts/finetune.py:698-705)
# keeps every parameter's gradient as a slice of a bucket; a gradient that arrives anywhere else is COPIED into its slice by the
# reducer's hook — 6.2 GB and ~260 launches per large-v3 step between the last dW GEMM of a bucket and its all-reduce.  The libwft
# optimizers note where each fp32 gradient lived at their step (`note_grad_homes`); the next backward's weight-gradient GEMM then
# writes (first micro-batch: param.grad is None, the returned alias is adopted by AccumulateGrad without a copy) or accumulates
# (later micro-batches under no_sync(): C += product in the reduce kernel, autograd gets None and adds nothing) straight into
# that memory.  The reducer finds grad.is_alias_of(bucket_view) and skips its copy.  Without DDP the homes are last step's
# gradient tensors: same code path, and gradient accumulation costs no elementwise add per parameter.  A home that no longer
# matches (DDP rebuilt its buckets after the first iteration, the user replaced .grad) is simply not used for that call.
# WFT_GRAD_HOMES=0 restores fresh gradient tensors per backward (A/B runs).
_GRAD_HOMES = os.environ.get("WFT_GRAD_HOMES", "1") != "0"


def note_grad_homes(params) -> None:
    if not _GRAD_HOMES:
        return
    for p in params:
        g = p.grad
        if g is not None and g.dtype == F32 and g.dim() == 2 and g.shape == p.shape and g.is_contiguous() and g.data_ptr() % 16 == 0 \
                and not p._backward_hooks:
            p.__dict__["_wft_grad_home"] = g


def _weight_grad_homes(weights, w_need):
    """-> (homes, accumulate) if EVERY weight of the group has a usable home and they agree on the mode, else None."""
    if not _GRAD_HOMES or not all(w_need):
        return None
    homes, modes = [], set()
    task = torch._C._current_graph_task_id()
    for w in weights:
        h = w.__dict__.get("_wft_grad_home")
        if h is None or h.shape != w.shape or h.device != w.device or w._backward_hooks:
            return None
        g = w.grad
        if g is None:
            # overwrite mode hands autograd an ALIAS of the home.  A second LinearFn node of the same weight in the same backward pass
            # (the model called twice before one backward, a Linear reused) still sees w.grad None — AccumulateGrad runs after both —
            # and would overwrite the first product: only the first writer of a graph task gets the home
            if w.__dict__.get("_wft_home_task") == task:
                return None
            modes.add(False)
        elif g.data_ptr() == h.data_ptr() and g.dtype == F32 and g.shape == h.shape and g.is_contiguous():
            modes.add(True)
        else:
            return None
        homes.append(h)
    if len(modes) != 1:
        return None
    mode = modes.pop()
    if not mode and task != -1:
        for w in weights:
            w.__dict__["_wft_home_task"] = task
    return homes, mode


class _LinearCfg:
    __slots__ = ("group", "n_w", "has_bias", "loras", "gelu_out", "gelu_in", "out_features", "accum", "gelu_next_n", "gelu_codes")

    def __init__(self, group, n_w, has_bias, loras, gelu_out, gelu_in, out_features, accum=None, gelu_next_n=0, gelu_codes=None):
        self.group, self.n_w, self.has_bias, self.loras = group, n_w, has_bias, loras
        self.gelu_out, self.gelu_in, self.out_features = gelu_out, gelu_in, out_features
        self.accum = accum
        self.gelu_next_n = gelu_next_n  # gelu_out: out-features of the Linear that consumes the activation (0: unknown -> bf16 gelu')
        self.gelu_codes = gelu_codes    # gelu_in: the one-byte gelu' buffer the producer returned beside `pre` (None: `pre` holds bf16 gelu')


# WFT_LORA_PVALID=0: rank-r weight-gradient GEMMs without the p_valid shortcut (A/B runs)
_LORA_PVALID = os.environ.get("WFT_LORA_PVALID", "1") != "0"
# WFT_LORA_FUSED_OUT=0: adapter gradients sliced / masked / transposed by torch ops after the GEMMs (A/B runs)
_LORA_FUSED_OUT = os.environ.get("WFT_LORA_FUSED_OUT", "1") != "0"
# WFT_LORA_PAIR=0: the four rank-r products of a group's backward as four launches + two reduces instead of two + one (A/B runs)
_LORA_PAIR = os.environ.get("WFT_LORA_PAIR", "1") != "0"
# WFT_GELU_PAIR=0: keep the pre-activation and evaluate gelu' in the backward-data GEMM's epilogue (A/B runs)
_GELU_PAIR = os.environ.get("WFT_GELU_PAIR", "1") != "0"
# WFT_GELU_AUX8=0: gelu' travels as bf16 [M, N] everywhere (A/B runs).  Default: ONE BYTE per element in gemm_nt4w_kernel's fragment
# order wherever both GEMMs of the pair run on that kernel (WFT_EPI_GELU_GRAD8 / WFT_EPI_MUL_AUX8, include/wft.h): half the bytes of
# the pair's second tensor, 21 GiB less saved-for-backward memory at 87 clips of large-v3.
_GELU_AUX8 = os.environ.get("WFT_GELU_AUX8", "1") != "0"


class LinearFn(torch.autograd.Function):
    """y = x @ Wcat^T + bias (+ LoRA) (GELU) (+ residual) as libwft GEMMs.

    inputs : x bf16 [M, K]; residual bf16 [M, N] or None; gelu_pre bf16 [M, K] or None
             (the first output of the gelu_out Linear that produced x: it stands for the pre-activation in the autograd
             graph — the backward returns d(pre) through it — but its VALUES are gelu'(pre), written by that GEMM's
             epilogue beside gelu(pre), so the backward-data GEMM here only multiplies by them: WFT_EPI_GELU_GRAD /
             WFT_EPI_MUL_AUX in include/wft.h);
             cfg; then weights..., biases (only the non-None ones)..., lora A..., lora B...
    outputs: y   (gelu_out=False)   |   (pre*, act) with act non-differentiable (gelu_out=True; pre* as above)
    """

    @staticmethod
    def forward(ctx, x, residual, gelu_pre, cfg: _LinearCfg, *params):
        ctx.set_materialize_grads(False)  # no [M, N] zero tensor for the non-differentiable `act` output
        n_w = cfg.n_w
        weights = list(params[:n_w])
        nb = sum(cfg.has_bias)
        bias_list = list(params[n_w:n_w + nb])
        biases, bi = [], 0
        for hb in cfg.has_bias:
            biases.append(bias_list[bi] if hb else None)
            bi += int(hb)
        need_dx = ctx.needs_input_grad[0] or (gelu_pre is not None and ctx.needs_input_grad[2])
        has_lora = any(s is not None for s in cfg.loras)
        W, WT, bias = cfg.group.shadows(weights, biases, need_t=need_dx, loras=cfg.loras if has_lora else None)
        n, k, npad = LinearGroup.dims(weights)
        kpad = k
        assert x.dtype == BF16 and x.dim() == 2 and x.is_contiguous() and x.shape[1] == kpad, (x.shape, kpad)
        M = x.shape[0]
        out_pre = aux8 = None
        if cfg.gelu_out:
            # one-byte gelu' if this GEMM AND the backward-data GEMM of the consuming Linear ([M, next_n] x [npad, next_n]^T with the
            # fused column sums) both run on gemm_nt4w_kernel
            nb8 = 0
            if _GELU_PAIR and _GELU_AUX8 and residual is None and cfg.gelu_next_n > 0:
                nb8 = K.gemm_nt_aux8_bytes(M, npad, kpad, x.device)
                if nb8 and not K.gemm_nt_aux8_bytes(M, npad, cfg.gelu_next_n, x.device, epilogue=L.EPI_MUL_AUX8, colsum=BIAS_GRADS[0]):
                    nb8 = 0
            if nb8:
                aux8 = torch.empty(nb8, dtype=torch.uint8, device=x.device)
                y = K.gemm_nt(x, W, bias=bias, epilogue=L.EPI_GELU_GRAD8, aux=aux8)
                # the stand-in for the pre-activation in the autograd graph: its gradient is a dense [M, npad] tensor, its VALUES
                # are never read (a zero-stride view of one element); the codes ride on the tensor object
                out_pre = torch.empty(1, dtype=BF16, device=x.device).expand(M, npad)
            else:
                out_pre = torch.empty((M, npad), dtype=BF16, device=x.device)
                y = K.gemm_nt(x, W, bias=bias, epilogue=L.EPI_GELU_GRAD if _GELU_PAIR else L.EPI_GELU, aux=out_pre, residual=residual)
        else:
            y = K.gemm_nt(x, W, bias=bias, residual=residual)
        ctx.cfg = cfg
        ctx.launch = _rt.backward_launch_mode()  # per-call launch mode of this node's backward GEMMs (runtime.exchange_launch_mode)
        ctx.accum = None
        if cfg.accum is not None and ctx.needs_input_grad[0] and gelu_pre is None:
            ctx.accum = cfg.accum
        ctx.has_res = residual is not None
        ctx.want_cs = BIAS_GRADS[0]
        ctx.dims = (n, k, npad, kpad)
        # (gelu_pre is either the bf16 gelu' matrix or the zero-stride stand-in whose one-byte codes arrive in cfg.gelu_codes)
        pre_codes = cfg.gelu_codes if gelu_pre is not None else None
        ctx.pre_is_aux8 = pre_codes is not None
        ctx.save_for_backward(x, pre_codes if pre_codes is not None else gelu_pre, None if aux8 is not None else out_pre, *params)
        if cfg.gelu_out:
            ctx.mark_non_differentiable(y)
            if aux8 is not None:
                ctx.mark_non_differentiable(aux8)
            return out_pre, y, aux8
        return y

    @staticmethod
    def backward(ctx, *grads):
        cfg: _LinearCfg = ctx.cfg
        x, gelu_pre, out_pre, *params = ctx.saved_tensors
        n, k, npad, kpad = ctx.dims
        n_w = cfg.n_w
        weights = list(params[:n_w])
        nb = sum(cfg.has_bias)
        has_lora = any(s is not None for s in cfg.loras)
        dy = grads[0]
        if dy is None:
            return (None,) * (4 + len(params))
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dy = dy.contiguous()
        W, WT, _ = cfg.group.shadows(weights, _bias_list(cfg, params), need_t=True, loras=cfg.loras if has_lora else None)
        # NB: with GELU-out the incoming grad is w.r.t. the pre-activation already (act is non-diff)
        dx = dpre = None
        need_dx = ctx.needs_input_grad[0] or (gelu_pre is not None and ctx.needs_input_grad[2])
        if need_dx:  # WT is the EFFECTIVE weight (base + adapters): one GEMM, as without LoRA
            if gelu_pre is not None:
                # dpre is the dy of the Linear that produced gelu_pre: its bias gradient (column sums) comes out of this
                # GEMM's epilogue (see _publish_colsum / _fused_colsum)
                cs = torch.empty(WT.shape[0], dtype=F32, device=dy.device) if ctx.want_cs else None
                epi = L.EPI_MUL_AUX8 if ctx.pre_is_aux8 else (L.EPI_MUL_AUX if _GELU_PAIR else L.EPI_DGELU)
                dpre = K.gemm_nt(dy, WT, epilogue=epi, aux=gelu_pre, colsum=cs, launch=ctx.launch)
                if cs is not None:
                    _publish_colsum(dpre, cs)
            elif ctx.accum is not None:  # one of several consumers of x: add into the running sum, the fork node returns it
                lm = ctx.launch
                dx = ctx.accum.arrive(lambda run: K.gemm_nt(dy, WT, launch=lm) if run is None
                                      else K.gemm_nt(dy, WT, residual=run, out=run, launch=lm))
            else:
                dx = K.gemm_nt(dy, WT, launch=ctx.launch)
        out: List[Optional[torch.Tensor]] = [dx, dy if ctx.has_res else None, dpre, None]
        # parameter grads: weights
        w_need = [ctx.needs_input_grad[4 + i] for i in range(n_w)]
        homed = _weight_grad_homes(weights, w_need) if (any(w_need) and k == kpad and n == npad) else None
        if homed is not None and K.gemm_tn(dy, x, seg_out=homed[0], accumulate=homed[1]) is not None:
            # each gradient written where it lives (a DDP bucket view): an alias for AccumulateGrad to adopt, or — accumulated in
            # place — nothing for it to add
            out.extend([None] * n_w if homed[1] else [h.detach() for h in homed[0]])
        elif any(w_need):
            dW = K.gemm_tn(dy, x)  # f32 [Npad, Kpad]
            off = 0
            for i, w in enumerate(weights):
                o = w.shape[0]
                out.append(dW[off:off + o, :k].reshape(w.shape) if w_need[i] else None)
                off += o
        else:
            out.extend([None] * n_w)
        # biases
        b_need = [ctx.needs_input_grad[4 + n_w + i] for i in range(nb)]
        if any(b_need):
            db = _fused_colsum(grads[0], dy)
            if db is None:
                db = K.colsum(dy)
            off = bi = 0
            for w, hb in zip(weights, cfg.has_bias):
                if hb:
                    out.append(db[off:off + w.shape[0]] if b_need[bi] else None)
                    bi += 1
                off += w.shape[0]
        else:
            out.extend([None] * nb)
        # LoRA A / B
        n_l = sum(1 for s in cfg.loras if s is not None)
        if has_lora:
            base = 4 + n_w + nb
            a_need = [ctx.needs_input_grad[base + i] for i in range(n_l)]
            b2_need = [ctx.needs_input_grad[base + n_l + i] for i in range(n_l)]
            Am, AmT, Bb, BbT = cfg.group.lora_shadows(weights, cfg.loras)
            dA_full = dB_full = None
            rtot = sum(s.A.shape[0] for s in cfg.loras if s is not None)
            # the rank-r operand of the weight-gradient GEMMs sits in a 128-wide zero-padded buffer: p_valid routes them to the
            # load-stream kernel for rank-r operands (35 us for the 123 MB activation of a 1280-wide Linear at 32 clips, 52 us
            # through the square-tile kernel); the two NT products below already run at the HBM rate of their activation
            # operand through the 128-wide tile kernel (27 us, measured — a dedicated skinny kernel was slower)
            pv = rtot if (_LORA_PVALID and rtot <= 64) else 0
            specs = [s for s in cfg.loras if s is not None]
            r0, n0 = specs[0].A.shape[0], weights[0].shape[0]
            # every Linear of the group adapted, equal shapes, nothing padded, every adapter gradient wanted (the normal case:
            # q/k/v, k/v or a single Linear): the kernel that sums the split-K partials applies the dropout masks to dA and
            # writes dB as [out, r] blocks, so both leave the library as the contiguous tensors autograd hands on — no
            # `dA * mask` kernel and no strided-view copy in AccumulateGrad per adapter (WFT_LORA_FUSED_OUT=0: the slicing below)
            fused = (_LORA_FUSED_OUT and pv > 0 and len(specs) == n_w and all(a_need) and all(b2_need) and kpad == k
                     and npad == n and all(s.A.shape[0] == r0 for s in specs) and all(w.shape[0] == n0 for w in weights)
                     and all((s.mask is None) == (specs[0].mask is None) for s in specs))
            if fused and _LORA_PAIR:
                # the four rank-r products as two paired launches (+ one reduce launch for both gradients): 3 launches, not 6
                du, u = K.gemm_nt_rank_pair(dy, BbT, x, Am, pv)          # dy @ (s*B) | x @ (s*A*mask)^T (carries the scaling)
                dA_full, dB_blocks = K.gemm_tn_rank_pair(
                    dict(a=du, b=x, p_valid=pv, col_scale=_stacked_masks(specs), scale_rows=r0 if n_w > 1 else 0),
                    dict(a=u, b=dy, p_valid=pv, block_n=n0, block_r=r0))
                out.extend(dA_full[i * r0:(i + 1) * r0] for i in range(n_w))
                out.extend(dB_blocks[i * n0 * r0:(i + 1) * n0 * r0].view(n0, r0) for i in range(n_w))
                return tuple(out)
            if fused:
                du = K.gemm_nt(dy, BbT, p_valid=pv)                      # [M, Rpad] = dy @ (s*B)
                dA_full = K.gemm_tn(du, x, p_valid=pv, col_scale=_stacked_masks(specs), scale_rows=r0 if n_w > 1 else 0)
                u = K.gemm_nt(x, Am, p_valid=pv)                       # [M, Rpad] = x @ (s*A*mask)^T (carries the scaling: wft_lora_pack)
                dB_blocks = K.gemm_tn(u, dy, p_valid=pv, block_n=n0, block_r=r0)
                out.extend(dA_full[i * r0:(i + 1) * r0] for i in range(n_w))
                out.extend(dB_blocks[i * n0 * r0:(i + 1) * n0 * r0].view(n0, r0) for i in range(n_w))
                return tuple(out)
            if any(a_need):
                du = K.gemm_nt(dy, BbT, p_valid=pv)                      # [M, Rpad] = dy @ (s*B)
                dA_full = K.gemm_tn(du, x, p_valid=pv)        # [Rpad, Kpad]
            if any(b2_need):
                u = K.gemm_nt(x, Am, p_valid=pv)                       # [M, Rpad] = x @ (s*A*mask)^T, recomputed here, not saved by the forward
                # rank-r operand first (its zero-padded columns are skipped): [Rpad, Npad], read through its transpose
                dB_full = K.gemm_tn(u, dy, p_valid=pv).t()    # [Npad, Rpad]
            dAs, dBs = [], []
            ro = no = li = 0
            for w, s in zip(weights, cfg.loras):
                if s is not None:
                    r = s.A.shape[0]
                    if a_need[li]:
                        g = dA_full[ro:ro + r, :k]
                        dAs.append(g * s.mask if s.mask is not None else g)
                    else:
                        dAs.append(None)
                    # u = x (s A*m)^T already carries the scaling (wft_lora_pack): dB is the slice itself
                    dBs.append(dB_full[no:no + w.shape[0], ro:ro + r] if b2_need[li] else None)
                    ro += r
                    li += 1
                no += w.shape[0]
            out.extend(dAs)
            out.extend(dBs)
        return tuple(out)


def _stacked_masks(specs):
    """The dropout masks of a group's adapters as one f32 [S, K] tensor (None without dropout): a view when they are
    consecutive slices of the model's mask pool (model/lora.py LoraMaskPool — q, k, v are drawn side by side), else a copy."""
    m0 = specs[0].mask
    if m0 is None:
        return None
    kk = m0.shape[-1]
    if len(specs) == 1:
        return m0.detach().view(1, kk)
    st = m0.untyped_storage().data_ptr()
    if all(s.mask.untyped_storage().data_ptr() == st and s.mask.is_contiguous()
           and s.mask.data_ptr() == m0.data_ptr() + 4 * kk * i for i, s in enumerate(specs)):
        return torch.as_strided(m0.detach(), (len(specs), kk), (kk, 1))
    return torch.cat([s.mask.detach().view(1, kk) for s in specs], 0)


# data_ptr -> (producing tensor, colsum), PER THREAD (round 6): producer and consumer are nodes of one backward pass, i.e. run on one
# autograd device thread; a forward on ANOTHER thread (an evaluator beside a training backward) used to clear the shared table
# mid-pass — the consumers then fell back to wft_colsum_bf16, whose sums differ from the fused ones in the last bit
# (tests/test_launch_mode_gpu.py failed one run in three on encoder.blocks.0.*.bias)
_COLSUMS_TLS = threading.local()


def _colsums() -> dict:
    d = getattr(_COLSUMS_TLS, "d", None)
    if d is None:
        d = _COLSUMS_TLS.d = {}
    return d


# Does any bias of the running model take a gradient?  Set by the model at the start of every forward (Whisper.forward /
# forward_loss).  False in a LoRA run (frozen base): the producer kernels then skip the fused bias-gradient column sums and
# their reduce launches (LayerNorm backward, attention backward, the DGELU / MUL_AUX GEMM epilogue).
BIAS_GRADS = [True]
# whisper's key projection has no bias; a model that gives it one (not whisper) must set this so that the fused q/k/v column
# sums (which leave the k slice at zero) are not used for it
_K_HAS_BIAS = [False]


def reset_colsums() -> None:
    """Drop this thread's unconsumed entries (called at the start of every forward and of every backward pass)."""
    _colsums().clear()


def _publish_colsum(t: torch.Tensor, cs: torch.Tensor) -> None:
    """Remember that `cs` holds the column sums of `t`, a gradient tensor a libwft kernel just wrote.  Autograd hands
    the consumer a VIEW of t (reshape nodes between modules), so the link is by address.  The entry keeps `t` itself
    alive until it is consumed or the next pass starts: its storage therefore cannot be recycled for other data, and a
    consumer tensor with the same address, storage and element count IS this data."""
    _colsums()[t.data_ptr()] = (t, cs)


def _fused_colsum(grad, dy):
    """Column sums a producer kernel already formed for exactly this gradient (LayerNorm backward for the residual
    stream, the DGELU GEMM epilogue for d(pre-activation)); None if autograd handed over different data (accumulated /
    cast / re-laid-out), in which case the caller falls back to wft_colsum_bf16."""
    ent = _colsums().pop(dy.data_ptr(), None)
    if ent is None:
        return None
    t, cs = ent
    if (grad.dtype != BF16 or t.numel() != dy.numel() or cs.numel() != dy.shape[-1] or not dy.is_contiguous()
            or t.untyped_storage().data_ptr() != dy.untyped_storage().data_ptr()):
        return None
    return cs


def _bias_list(cfg, params):
    n_w = cfg.n_w
    bl = list(params[n_w:n_w + sum(cfg.has_bias)])
    res, bi = [], 0
    for hb in cfg.has_bias:
        res.append(bl[bi] if hb else None)
        bi += int(hb)
    return res


def linear(x, group: LinearGroup, weights, biases, loras=None, residual=None, gelu_out=False, gelu_pre=None, dx_accum=None, gelu_next_n=0,
           gelu_codes=None):
    """Functional front door of LinearFn. x bf16 [M, K] contiguous.  dx_accum: the GradAccum of `grad_fork(x)` (x must be that fork).
    gelu_out -> (pre, act, codes): `pre` stands for the pre-activation in the autograd graph (its values are gelu'(pre) in bf16, or —
    codes is not None — nothing at all: a zero-stride stand-in, the derivative travels as one byte per element in `codes`); hand both to the
    consuming Linear as gelu_pre / gelu_codes.  gelu_next_n (with gelu_out): out-features of that consumer — lets the pair use the
    one-byte form when both of its GEMMs qualify."""
    loras = list(loras) if loras is not None else [None] * len(weights)
    cfg = _LinearCfg(group, len(weights), tuple(b is not None for b in biases), tuple(loras), gelu_out,
                     gelu_pre is not None, sum(w.shape[0] for w in weights), dx_accum, int(gelu_next_n), gelu_codes)
    params = list(weights) + [b for b in biases if b is not None]
    params += [s.A for s in loras if s is not None] + [s.B for s in loras if s is not None]
    return LinearFn.apply(x, residual, gelu_pre, cfg, *params)


# --------------------------------------------------------------------------- stochastic depth
class SdRescaleFn(torch.autograd.Function):
    """x + (out - x) / keep in one kernel (StochasticDepthMixin.stochastic_depth, model/model_utils.py:241-250)."""

    @staticmethod
    def forward(ctx, x, out, keep: float):
        s = 1.0 / keep
        ctx.s = s
        return K.axpby_bf16(1.0 - s, x, s, out)

    @staticmethod
    def backward(ctx, g):
        g = g.to(BF16)
        return K.axpby_bf16(1.0 - ctx.s, g), K.axpby_bf16(ctx.s, g), None


# --------------------------------------------------------------------------- LayerNorm
class LayerNormFn(torch.autograd.Function):
    """whisper.model.LayerNorm (fp32 statistics, bf16 in/out) + optional deep-SpecAugment mask
    (rows_per_batch, t0, t1, c0, c1) fused in the same pass (model/model_utils.py:409-417)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, mask):
        shape = x.shape
        y, mean, rstd = K.layernorm_fwd(x.reshape(-1, shape[-1]), gamma.detach(), beta.detach(), eps, mask)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.mask = mask
        ctx.want_cs = BIAS_GRADS[0]
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        shape = x.shape
        want_cs, want_p = ctx.want_cs, ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        res = K.layernorm_bwd(dy.reshape(-1, shape[-1]).to(BF16), x.reshape(-1, shape[-1]), gamma.detach(), mean,
                              rstd, None, ctx.mask, want_colsum=want_cs, want_params=want_p)
        dx, dg, db = res[0].view(shape), res[1], res[2]
        if want_cs:
            _publish_colsum(dx, res[3])  # dx is the dy of the Linear that wrote x: its bias gradient, for free
        return dx, dg, db, None, None


class LayerNormForkFn(torch.autograd.Function):
    """(ln(x), x): the residual stream forks here; the backward fuses the add of the residual
    gradient into the LayerNorm backward kernel (one pass instead of LN-bwd + add)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, mask):
        ctx.set_materialize_grads(False)  # a missing branch gradient arrives as None, not as a zero tensor
        shape = x.shape
        y, mean, rstd = K.layernorm_fwd(x.reshape(-1, shape[-1]), gamma.detach(), beta.detach(), eps, mask)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.mask = mask
        ctx.want_cs = BIAS_GRADS[0]
        return y.view(shape), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, mean, rstd = ctx.saved_tensors
        shape = x.shape
        if dy is None:
            return dres, None, None, None, None
        dr = None if dres is None else dres.reshape(-1, shape[-1]).to(BF16)
        want_cs, want_p = ctx.want_cs, ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        res = K.layernorm_bwd(dy.reshape(-1, shape[-1]).to(BF16), x.reshape(-1, shape[-1]), gamma.detach(), mean,
                              rstd, dr, ctx.mask, want_colsum=want_cs, want_params=want_p)
        dx, dg, db = res[0].view(shape), res[1], res[2]
        if want_cs:
            _publish_colsum(dx, res[3])
        return dx, dg, db, None, None


# --------------------------------------------------------------------------- attention
class SelfAttnFn(torch.autograd.Function):
    """qkv bf16 [B, T, 3*d] (fused projection output, consumed in place) -> o bf16 [B, T, d]."""

    @staticmethod
    def forward(ctx, qkv, n_head, causal, prescaled=False):
        """prescaled: the q third of qkv carries scale * log2(e) (QK_PRESCALE: the projection's LinearGroup has fwd_scales)."""
        d = qkv.shape[-1] // 3
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        scale = 64 ** -0.5
        o, lse = K.attn_fwd(q, k, v, n_head, causal, scale, q_prescaled=prescaled)
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (n_head, causal, scale, bool(prescaled), _rt.backward_launch_mode())
        ctx.want_cs = BIAS_GRADS[0]
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        n_head, causal, scale, prescaled, launch = ctx.cfg
        d = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        want_cs = ctx.want_cs
        cs = torch.zeros(3 * d, dtype=F32, device=qkv.device) if want_cs else None  # [q | k (no bias in whisper: stays 0) | v]
        K.attn_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o, lse, do.to(BF16), n_head, causal, scale,
                   dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], colsums=(cs[:d], cs[2 * d:]) if want_cs else None,
                   q_prescaled=prescaled, launch=launch)
        if want_cs and not _K_HAS_BIAS[0]:
            _publish_colsum(dqkv, cs)  # bias gradients of the fused q/k/v projection, summed in the kernels' epilogues
        return dqkv, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """q bf16 [B, S, d], kv bf16 [B, T, 2*d] -> o bf16 [B, S, d] (no mask)."""

    @staticmethod
    def forward(ctx, q, kv, n_head):
        d = q.shape[-1]
        scale = 64 ** -0.5
        o, lse = K.attn_fwd(q, kv[..., :d], kv[..., d:], n_head, False, scale)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.cfg = (n_head, scale, _rt.backward_launch_mode())
        ctx.want_cs = BIAS_GRADS[0]
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        n_head, scale, launch = ctx.cfg
        d = q.shape[-1]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        want_cs = ctx.want_cs
        cs_q = torch.empty(d, dtype=F32, device=q.device) if want_cs else None
        cs_kv = torch.zeros(2 * d, dtype=F32, device=q.device) if want_cs else None  # [k (no bias) | v]
        K.attn_bwd(q, kv[..., :d], kv[..., d:], o, lse, do.to(BF16), n_head, False, scale, dq=dq, dk=dkv[..., :d], dv=dkv[..., d:],
                   colsums=(cs_q, cs_kv[d:]) if want_cs else None, launch=launch)
        if want_cs:
            _publish_colsum(dq, cs_q)
            if not _K_HAS_BIAS[0]:
                _publish_colsum(dkv, cs_kv)
        return dq, dkv, None


# --------------------------------------------------------------------------- conv stem
class ConvStemFn(torch.autograd.Function):
    """AudioEncoder stem (model/model_utils.py:276-281): gelu(conv1(mel)) -> gelu(conv2(.)) ->
    permute -> + positional_embedding, as im2col-free GEMMs over a time-major, zero-haloed layout.

    mel_t bf16 [B, T+2, c_pad] (wft_mel_to_tmajor_bf16); w1 f32 [d, n_mels, 3]; w2 f32 [d, d, 3];
    pos f32 [T/2, d] (a buffer)  ->  x bf16 [B, T/2, d].
    """

    @staticmethod
    def forward(ctx, mel_t, w1, b1, w2, b2, pos, cache):
        B, Tp, c_pad = mel_t.shape
        T = Tp - 2
        d = w1.shape[0]
        dev = mel_t.device
        key = (_ver(w1), _ver(w2), _ver(pos), _SHADOW_EPOCH[0])
        if cache.get("key") != key:
            note_shadowed((w1, w2))
            n_mels = w1.shape[1]
            w1p = torch.zeros((d, 3, c_pad), dtype=F32, device=dev)
            w1p[:, :, :n_mels] = w1.detach().permute(0, 2, 1)
            cache["w1"] = K.cast_bf16(w1p.reshape(d, 3 * c_pad))
            w2k = w2.detach().permute(0, 2, 1).contiguous()  # [co, kk, ci]
            cache["w2"] = K.cast_bf16(w2k.reshape(d, 3 * d))
            # backward-data shadows: odd rows use W2[:,:,1]^T, even rows [W2[:,:,2]^T | W2[:,:,0]^T]
            cache["w2_odd"] = K.cast_bf16(w2.detach()[:, :, 1].t().contiguous())  # [ci, co]
            cache["w2_even"] = K.cast_bf16(torch.cat([w2.detach()[:, :, 2].t(), w2.detach()[:, :, 0].t()], dim=1).contiguous())  # [ci, 2*co]
            cache["pos"] = K.cast_bf16(pos.detach().contiguous())
            cache["key"] = key
        T2 = T // 2
        pre1 = torch.empty((B, Tp, d), dtype=BF16, device=dev)
        act1 = torch.empty((B, Tp, d), dtype=BF16, device=dev)
        act1[:, 0].zero_(); act1[:, T + 1].zero_()
        K.gemm_nt(mel_t, cache["w1"], M=T, N=d, K=3 * c_pad, lda=c_pad, ldb=3 * c_pad, out=act1[:, 1:], ldc=d,
                  bias=b1.detach(), epilogue=L.EPI_GELU, aux=pre1[:, 1:], batch=B, strideA=Tp * c_pad, strideC=Tp * d,
                  strideAux=Tp * d)
        pre2 = torch.empty((B, T2, d), dtype=BF16, device=dev)
        x = torch.empty((B, T2, d), dtype=BF16, device=dev)
        K.gemm_nt(act1, cache["w2"], M=T2, N=d, K=3 * d, lda=2 * d, ldb=3 * d, out=x, ldc=d, bias=b2.detach(),
                  epilogue=L.EPI_GELU, aux=pre2, residual=cache["pos"], batch=B, strideA=Tp * d, strideC=T2 * d,
                  strideAux=T2 * d, strideR=0)
        ctx.save_for_backward(mel_t, pre1, act1, pre2)
        ctx.cache = cache
        ctx.shapes = (B, T, c_pad, d, w1.shape[1])
        ctx.launch = _rt.backward_launch_mode()
        return x

    @staticmethod
    def backward(ctx, dx):
        mel_t, pre1, act1, pre2 = ctx.saved_tensors
        cache = ctx.cache
        B, T, c_pad, d, n_mels = ctx.shapes
        Tp, T2 = T + 2, T // 2
        dev = dx.device
        dx = dx.to(BF16).contiguous()
        # through gelu of conv2 (positional embedding is a buffer: no grad)
        dpre2 = torch.zeros((B, T2 + 2, d), dtype=BF16, device=dev)  # zero halo rows 0 and T2+1
        tmp = K.dgelu_mul(dx.view(-1), pre2.view(-1)).view(B, T2, d)
        dpre2[:, 1:T2 + 1] = tmp
        db2 = K.colsum(tmp.view(B * T2, d))
        # dW2[co, kk*d + ci] = sum_{b,t'} dpre2[b,t',co] * act1pad[b, 2t'+kk, ci]
        dW2 = K.gemm_tn(dpre2[:, 1:], act1, R=T2, P=d, Q=3 * d, lda=d, ldb=2 * d, batch=B,
                        strideA=(T2 + 2) * d, strideB=Tp * d)
        dW2 = dW2.view(d, 3, d).permute(0, 2, 1)  # -> [co, ci, kk]
        # backward-data into the padded conv1 activation, fused with gelu'(pre1)
        dpre1 = torch.zeros((B, Tp, d), dtype=BF16, device=dev)
        # odd padded rows tau = 2j+1 (j = 0..T2-1): dpre2[j] @ W2[:,:,1]
        K.gemm_nt(dpre2[:, 1:], cache["w2_odd"], M=T2, N=d, K=d, lda=d, ldb=d, out=dpre1[:, 1:], ldc=2 * d,
                  epilogue=L.EPI_DGELU, aux=pre1[:, 1:], ldaux=2 * d, batch=B, strideA=(T2 + 2) * d, strideC=Tp * d,
                  strideAux=Tp * d, launch=ctx.launch)
        # even padded rows tau = 2j (j = 1..T2): [dpre2[j-1], dpre2[j]] @ [W2[:,:,2]; W2[:,:,0]]
        aux_even = pre1[:, 2:]
        K.gemm_nt(dpre2[:, 1:], cache["w2_even"], M=T2, N=d, K=2 * d, lda=d, ldb=2 * d, out=dpre1[:, 2:], ldc=2 * d,
                  epilogue=L.EPI_DGELU, aux=aux_even, ldaux=2 * d, batch=B, strideA=(T2 + 2) * d, strideC=Tp * d,
                  strideAux=Tp * d, launch=ctx.launch)
        db1 = K.colsum(dpre1.view(B * Tp, d))
        dW1 = K.gemm_tn(dpre1[:, 1:], mel_t, R=T, P=d, Q=3 * c_pad, lda=d, ldb=c_pad, batch=B, strideA=Tp * d,
                        strideB=Tp * c_pad)
        dW1 = dW1.view(d, 3, c_pad)[:, :, :n_mels].permute(0, 2, 1)
        return None, dW1, db1, dW2, db2, None, None


# --------------------------------------------------------------------------- embedding / logits / loss
class EmbedFn(torch.autograd.Function):
    """token_embedding(tokens) + positional_embedding[:S] -> bf16 (model/model_utils.py:317-318)."""

    @staticmethod
    def forward(ctx, tokens, emb, pos):
        out = K.embed_fwd(tokens, emb.detach(), pos.detach())
        ctx.save_for_backward(tokens)
        ctx.shapes = (emb.shape, pos.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        es, ps = ctx.shapes
        demb = torch.zeros(es, dtype=F32, device=dout.device)
        dpos = torch.zeros(ps, dtype=F32, device=dout.device)
        K.embed_bwd(tokens, dout.to(BF16), demb, dpos)
        return None, demb, dpos


class TiedLogitsFn(torch.autograd.Function):
    """logits = x @ E^T (tied output projection, model/model_utils.py:325), bf16 [M, Vpad]."""

    @staticmethod
    def forward(ctx, x, emb, group: LinearGroup):
        W, WT, _ = group.shadows([emb], [None], need_t=True)
        logits = K.gemm_nt(x, W)
        ctx.save_for_backward(x, emb)
        ctx.group = group
        ctx.launch = _rt.backward_launch_mode()
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, emb = ctx.saved_tensors
        W, WT, _ = ctx.group.shadows([emb], [None], need_t=True)
        dl = dl.to(BF16).contiguous()
        dx = K.gemm_nt(dl, WT, launch=ctx.launch) if ctx.needs_input_grad[0] else None
        dE = K.gemm_tn(dl, x)[: emb.shape[0], : emb.shape[1]] if ctx.needs_input_grad[1] else None
        return dx, dE, None


class FusedCEFn(torch.autograd.Function):
    """mean label-smoothed CE over non-ignored targets from padded bf16 logits; the backward
    overwrites the logits buffer with dlogits (no second [M, V] tensor)."""

    @staticmethod
    def forward(ctx, logits, targets, V, eps):
        row_loss, row_lse, stats, _ = K.ce_fwd(logits, targets, V, eps)
        ctx.save_for_backward(logits, targets, row_lse, stats)
        ctx.cfg = (V, eps)
        return stats[0] / stats[1]

    @staticmethod
    def backward(ctx, g):
        reset_colsums()  # a new backward pass starts here
        logits, targets, row_lse, stats = ctx.saved_tensors
        V, eps = ctx.cfg
        dl = K.ce_bwd(logits, targets, V, eps, row_lse, stats, g, inplace=True)
        return dl, None, None, None
