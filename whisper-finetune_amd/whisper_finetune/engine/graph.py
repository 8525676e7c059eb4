"""HIP-graph capture of one micro-batch (forward + fused loss + backward) of the engine model.

Why: the small configurations of the reference (BASELINE.json configs[1]: whisper-base, 8 clips) are launch-bound — 573 kernel
launches for 11.9 ms of kernels per optimizer step, 14-16 ms per step with the Python autograd functions and the ctypes calls in
between (profiles/r05_base_kernel_stats.txt).  The reference's loop (scripts/finetune.py:177-188 -> model_utils.train_step) calls the
model once per micro-batch with tensors of ONE shape per (batch size, decoder length); captured once, that call is a single graph
launch.  `training.wft_hip_graph: true` makes `model_utils.train_step` route its micro-batches through `GraphedMicroBatch`.

What is captured: autocast + model(x, y_in, targets=y_out, label_smoothing) / accum + backward(), on static input buffers, with
every trainable parameter's .grad a PERSISTENT slice of one flat buffer that the graph's last node (one multi-tensor add over the
graph-private gradient tensors its AccumulateGrad nodes adopted) adds into — all graphs (one per input shape) share those buffers
and gradient accumulation needs no second graph; the step zeroes the flat buffer in place (one fill) instead of dropping the
gradients.  What is not: the GPU front end (log-mel + SpecAugment: its spans are host draws), clipping, the
optimizer and the scheduler (learning rate and bias corrections are kernel arguments that change every step).

Everything whose kernel ARGUMENTS are drawn on the host per call freezes under capture, so the wrapper refuses (loudly, once, and
train_step falls back to the eager path): stochastic depth, deep SpecAugment hooks, LoRA dropout, DDP wrappers (their reducer
hooks are Python), the fp32 compute mode's host-side paths are fine but untested here and refused too.
Results are the eager path's bit for bit (same kernels, same order): tests/test_hip_graph_gpu.py."""
from __future__ import annotations

import weakref
from typing import Optional

import torch

# model -> (key, GraphedMicroBatch).  Kept OFF the module: a captured graph holds torch.cuda.CUDAGraph objects, which neither pickle nor
# deep-copy, and `save_model` deep-copies the model (reference model/model_utils.py:130-135).  Weak keys: a dropped model drops its graphs.
_GRAPHS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
MAX_SHAPES = 8  # captured input shapes per model; further shapes run eagerly (each graph pins its static inputs and pool memory)


def graphed_for(model):
    """The (key, GraphedMicroBatch) entry of `model`, or None."""
    return _GRAPHS.get(model)


def has_graphs(model) -> bool:
    return model in _GRAPHS


def set_graphed(model, key, gm) -> None:
    _GRAPHS[model] = (key, gm)


def why_not(model) -> Optional[str]:
    """None if `model` (un-wrapped engine Whisper) can run its micro-batches from a captured graph, else the reason."""
    from whisper_finetune.engine.whisper_model import Whisper

    if not isinstance(model, Whisper):
        return "not an engine Whisper model (DDP wrappers run Python hooks in the backward pass)"
    params = [p for p in model.parameters()]
    if not params or not params[0].is_cuda:
        return "the model is not on a HIP device"
    if getattr(model, "compute_dtype", "bf16") != "bf16":
        return "only the bf16 compute mode is captured"
    for part in (model.encoder, model.decoder):
        if getattr(part, "stochastic_depth_prob", 0.0) > 0.0:
            return "stochastic depth draws its skips on the host per call"
    if "_wft_lora_pool" in model.__dict__:
        pool = model.__dict__["_wft_lora_pool"]
        if any(getattr(a, "lora_dropout_p", 0.0) > 0.0 for a in pool.adapters):
            return "LoRA dropout masks are redrawn per forward"
    for m in model.modules():
        if getattr(m, "deep_spec_augment", None) is not None:
            return "deep SpecAugment draws its spans on the host per call"
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return "multi-process job: gradient exchange happens in Python hooks"
    return None


class GraphedMicroBatch:
    def __init__(self, model, label_smoothing: float, accum: int, amp_dtype=torch.bfloat16, warmup: int = 2):
        self._model = weakref.ref(model)  # (the registry's key must not be kept alive by its value)
        self.ls, self.accum, self.amp_dtype, self.warmup = float(label_smoothing), int(accum), amp_dtype, warmup
        self.graphs = {}  # input shapes -> (graph, x, y_in, y_out, loss)
        self.pool = None
        self.eager_calls = 0
        self.eager_after_step = 0  # eager micro-batches that ran behind an optimizer step (the batched shadow refresh builds its
        #                            pointer table in the first of them: a host-to-device copy, not allowed under capture)
        self.disabled = None       # the reason, if a capture failed: eager from then on
        from whisper_finetune.engine import ops

        self._epoch0 = ops._SHADOW_EPOCH[0]
        self.device = next(model.parameters()).device
        # persistent gradient buffers (see the module docstring).  Round 6: slices of ONE flat fp32 buffer — the step zeroes it with
        # one fill instead of one per parameter (245 launches per whisper-base step) — and a captured backward no longer adds into
        # them parameter by parameter: see _capture
        self.flat = None
        self._views = []  # (parameter, its slice of self.flat)
        train = [p for p in model.parameters() if p.requires_grad]
        if train and all(p.dtype == torch.float32 and p.is_contiguous() for p in train):
            offs, total = [], 0
            for p in train:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64  # 256-byte slices: 16-byte vector accesses in the optimizer kernels
            self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            for p, o in zip(train, offs):
                v = self.flat[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
                self._views.append((p, v))
        else:
            for p in train:
                if p.grad is None:
                    p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)

    @property
    def model(self):
        m = self._model()
        if m is None:
            raise RuntimeError("the model of this GraphedMicroBatch no longer exists")
        return m

    def zero_grads(self, optimizer) -> None:
        """The step's `zero_grad(set_to_none=False)`: one fill of the flat buffer while every gradient still is its slice of it."""
        if self.flat is not None and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in self._views):
            self.flat.zero_()
        else:
            optimizer.zero_grad(set_to_none=False)

    def _eager(self, x, y_in, y_out):
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
            loss = self.model(x, y_in, targets=y_out, label_smoothing=self.ls) / self.accum
        loss.backward()
        return loss.detach()

    def __call__(self, x, y_in, y_out) -> torch.Tensor:
        """Runs forward + backward of one micro-batch (gradients ADD into the persistent .grad buffers); returns the scaled loss as a
        device scalar."""
        from whisper_finetune.engine import ops

        key = (tuple(x.shape), x.dtype, tuple(y_in.shape), tuple(y_out.shape))
        # a replay adds into the persistent slices whatever .grad says: gradients dropped meanwhile (an evaluation loop's
        # zero_grad(set_to_none=True), a checkpoint reload) get their zeroed slice back — the optimizer reads .grad
        for p, v in self._views:
            if p.grad is None:
                v.zero_()
                p.grad = v
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= MAX_SHAPES and self.disabled is None:
                self.shapes_refused = getattr(self, "shapes_refused", 0) + 1
                if self.shapes_refused == 1:
                    print(f"WARNING: {MAX_SHAPES} input shapes are captured already; further shapes run eagerly "
                          "(pad decoder lengths to a small set to keep them on the graph path).")
                return self._eager(x, y_in, y_out)
            if self.disabled is not None or self.eager_calls < self.warmup or self.eager_after_step < 1:
                # the first calls run eagerly: lazy one-time work (kernel attributes, workspaces, bf16 shadows, pointer tables) must
                # not happen under capture — at least two micro-batches, one of them behind an optimizer step
                self.eager_calls += 1
                if ops._SHADOW_EPOCH[0] != self._epoch0:
                    self.eager_after_step += 1
                return self._eager(x, y_in, y_out)
            try:
                ent = self._capture(key, x, y_in, y_out)
            except RuntimeError as err:  # nothing has executed under capture: say so and stay eager
                self.disabled = str(err).splitlines()[0]
                print(f"WARNING: HIP-graph capture of the micro-batch failed ({self.disabled}); this run stays on the eager path.")
                return self._eager(x, y_in, y_out)
        g, sx, sy_in, sy_out, sloss = ent
        sx.copy_(x, non_blocking=True)
        sy_in.copy_(y_in, non_blocking=True)
        sy_out.copy_(y_out, non_blocking=True)
        g.replay()
        return sloss

    def _capture(self, key, x, y_in, y_out):
        from whisper_finetune.engine import ops

        # the bf16 weight shadows are refreshed by the first forward behind an optimizer step (a Python-side staleness check that a
        # replay does not repeat): make them stale NOW so that the refresh launch is part of every graph, whatever micro-batch
        # happens to be the one that is captured
        ops.bump_shadow_epoch()
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        sx, sy_in, sy_out = x.clone(), y_in.clone(), y_out.clone()
        g = torch.cuda.CUDAGraph()
        # A captured AccumulateGrad node that finds a .grad ADDS into it: one elementwise launch per parameter (243 of the 830 launches
        # of a whisper-base micro-batch, 1.2 ms of its 11.6 ms of kernels).  Parameters are therefore captured WITHOUT a .grad — the
        # node adopts the incoming gradient tensor (graph-private memory, no kernel) — and ONE multi-tensor add at the end of the graph
        # folds those tensors into the persistent buffers.  Large 2-D weights whose weight-gradient GEMM can accumulate straight into
        # its gradient home (ops._weight_grad_homes: the persistent buffer itself; served by the 256 x 256 kernels only) keep their
        # .grad: nothing is added for them either way.  The others lose their home for the capture: a weight without a .grad but with
        # a home would be OVERWRITTEN in it (the first-micro-batch mode of the eager path), not accumulated.
        params = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        kept, homes = [], []
        for p in params:
            h = p.__dict__.get("_wft_grad_home")
            if h is not None and h.data_ptr() == p.grad.data_ptr() and p.dim() == 2 and min(p.shape) >= 1024 \
                    and p.shape[0] % 256 == 0 and p.shape[1] % 256 == 0:
                continue
            if h is not None:
                homes.append((p, p.__dict__.pop("_wft_grad_home")))
            kept.append((p, p.grad))
            p.grad = None
        try:
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, pool=self.pool, stream=side):
                    with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
                        loss = self.model(sx, sy_in, targets=sy_out, label_smoothing=self.ls) / self.accum
                    loss.backward()
                    sloss = loss.detach()
                    dst = [pg for p, pg in kept if p.grad is not None]
                    src = [p.grad for p, pg in kept if p.grad is not None]
                    if dst:
                        torch._foreach_add_(dst, src)
                    del src
        finally:
            for p, pg in kept:
                p.grad = pg
            for p, h in homes:
                p.__dict__["_wft_grad_home"] = h
        cur.wait_stream(side)
        if self.pool is None:
            self.pool = g.pool()
        ent = self.graphs[key] = (g, sx, sy_in, sy_out, sloss)
        return ent
