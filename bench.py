#!/usr/bin/env python
"""bench.py — audio-seconds/sec training throughput, whisper-large-v3 bf16 (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimizer step of the hot path on every rank: synthetic 30 s clips already
resident in HBM -> log-mel -> SpecAugment -> encoder/decoder forward -> label-smoothed CE ->
backward -> (DDP gradient all-reduce over RCCL, overlapped with backward) -> grad-norm clip folded
into the libwft multi-tensor AdamW — driven by the product's own `model_utils.train_step` (per-micro-batch loss.item(),
scheduler step included).  68 clips per GPU per step by default (180 GiB of the 288 GB HBM).
Weights: random init of the whisper-large-v3 architecture; data: synthetic (no network).
Rank 0 prints ONE JSON line (contract in the task statement); it also carries
  "roofline":     the dominant kernel (the NT GEMM family with the most time — gemm_nt4w_kernel since round 4: the
                  Linear / logits forward and backward-data GEMMs), algorithmic FLOPs (2*M*N*K) / HIP-event time of its launches
                  during one instrumented step that follows the timed region; "traffic" = HBM-side
                  bytes per launch from the committed rocprofv3 PMC passes of this command (profiles/);
  "cpu_baseline": the CPU oracle (oracle/whisper_oracle.py, torch fp32) forward+backward on a
                  bounded sample, timed on this box's host cores (N=1 only);
and, on the default (headline) invocation, extra keys measured in the same process after the headline:
  "hand_rolled_ms_per_step": round 1's hand-written loop on the same state (cross-check of the product loop),
  "reference_yaml_shapes":   the reference YAML's batch 32 at S = 128 and at the S = 448 stress length,
  "configs2_lora_muon":      BASELINE configs[2] (LoRA r16 + Muon + stochastic depth + deep SpecAugment, B = 32) with
                             its own roofline object,
  "configs1_base":           BASELINE configs[1] (whisper-base full fine-tune, B = 8, S = 128) with training.wft_hip_graph (the micro-batch
                             as one captured HIP graph); the eager step time is in the same object,
  "configs4_turbo_lora":     BASELINE configs[4]'s per-GPU workload (large-v3-turbo, LoRA r16, 64-token prompt with -100 targets +
                             timestamp tokens every 16th position, B = 64 — SURVEY.md §8d config 5),
  "entrypoint_loop":         scripts/finetune.py's real loop (SyntheticDataset -> DataLoader workers -> GpuMelLoader ->
                             train_step, tools/e2e_entrypoint.py) at B = 32 next to the same batches replayed from HBM.
  "ddp_mode_1gpu":           the headline re-timed in the WORLD_SIZE > 1 launch configuration on this one GPU (DDP wrapper + per-tile
                             launches), with and without a side-stream kernel that holds CUs and moves the ring's bytes,
  "cpu_baseline_small_models": the oracle's step for whisper-tiny / whisper-base (B = 2) on all host cores and on one.
`ms_per_step` is the mean over the K timed steps (the contract's clock); `ms_per_step_median` is the median of the per-step
wall times inside that region (train_step ends with loss.item()).  `--no-extras` prints the headline only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def fwd_flops_per_clip(d, S: int) -> float:
    """SURVEY.md §8d algorithmic forward FLOPs per 30 s clip."""
    T, dm, V = d.n_audio_ctx, d.n_audio_state, d.n_vocab
    conv = 2 * d.n_mels * dm * 3 * 3000 + 2 * dm * dm * 3 * 1500
    enc = d.n_audio_layer * (24 * T * dm * dm + 4 * T * T * dm)
    dec = d.n_text_layer * (28 * S * dm * dm + 4 * T * dm * dm + 4 * S * S * dm + 4 * S * T * dm)
    return conv + enc + dec + 2 * S * dm * V


def attn_core_flops_per_clip(d, S: int) -> float:
    T, dm = d.n_audio_ctx, d.n_audio_state
    return d.n_audio_layer * 4 * T * T * dm + d.n_text_layer * (4 * S * S * dm + 4 * S * T * dm)


def lora_flops_per_clip(d, S: int, r: int) -> float:
    """F_lora = sum over adapted Linears of 2*M*r*(in+out)  (SURVEY.md §8d)."""
    T, dm = d.n_audio_ctx, d.n_audio_state
    per_row_attn = 4 * 2 * r * (dm + dm)            # q, k, v, out
    per_row_mlp = 2 * r * (dm + 4 * dm) * 2         # mlp.0, mlp.2
    enc = d.n_audio_layer * T * (per_row_attn + per_row_mlp)
    dec = d.n_text_layer * (S * (per_row_attn + per_row_mlp) + S * 2 * 2 * r * 2 * dm + T * 2 * 2 * r * 2 * dm)  # cross q,out on S; k,v on T
    return enc + dec


def pmc_traffic_per_launch(batch: int, lora: bool = False, kname: str = "gemm_nt256_kernel"):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/collect_r06.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, the only tracing next to them
    being --kernel-trace).  FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 for 16-B/lane streaming
    reads, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KiB) is exact.  Counters cannot be
    collected inside the timed run, so the value is only reported when the committed pass used the same batch."""
    cands = sorted((ROOT / "profiles").glob("r0*_lora_pmc_summary.json" if lora else "r0*_pmc_summary.json"))
    if not lora:
        cands = [c for c in cands if "_lora_" not in c.name]
    if not cands:
        return None, "no PMC summary committed"
    path = cands[-1]  # the latest round's pass
    d = json.loads(path.read_text())
    if d.get("batch", 68) != batch:
        return None, f"PMC pass was collected at batch {d.get('batch', 68)}"
    tot = n = 0.0
    for k, v in d["FETCH_SIZE"].items():
        if kname in k:
            w = d["WRITE_SIZE"][k]
            tot += (2.0 * v["sum"] + w["sum"]) * 1024.0
            n += v["launches"]
    if n == 0:
        return None, "kernel not in the PMC summary"
    return round(tot / n), f"profiles/{path.name}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes"


def build_model(name: str, device, sd_p: float = 0.0):
    from whisper_finetune.engine.whisper_model import MODEL_DIMS, Whisper, sinusoids

    dims = MODEL_DIMS[name]
    torch.manual_seed(0)
    with torch.device(device):
        model = Whisper(dims)
        if sd_p > 0:
            from whisper_finetune.model.model_utils import CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder

            model.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head,
                                                               dims.n_audio_layer, sd_p)
            model.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                              dims.n_text_layer, sd_p)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
        model.encoder.positional_embedding.copy_(sinusoids(dims.n_audio_ctx, dims.n_audio_state))
    return model, dims


def synthetic_tokens(B: int, S: int, device, seed: int, prompt_ts: bool = False):
    """y_in = [sot, <|de|>, <|transcribe|>, <|notimestamps|>] + U{0..50256}^(S-4), y_out = shift + [eot].
    prompt_ts (config 5): [sot_prev] + a 64-token prompt with -100 targets in front, no <|notimestamps|>, and a timestamp token
    timestamp_begin + U{0..1500} at every 16th position of the transcript."""
    g = torch.Generator().manual_seed(4321 + seed)
    if not prompt_ts:
        specials = torch.tensor([50258, 50261, 50359, 50363])
        body = torch.randint(0, 50257, (B, S - 4), generator=g)
        y_in = torch.cat([specials.expand(B, -1), body], dim=1)
        y_out = torch.cat([y_in[:, 1:], torch.full((B, 1), 50257)], dim=1)
        return y_in.to(device), y_out.to(device)
    n_prompt = 64
    body = torch.randint(0, 50257, (B, S - 3), generator=g)
    ts = 50364 + torch.randint(0, 1501, (B, S - 3), generator=g)
    pos = torch.arange(S - 3)
    body = torch.where((pos % 16 == 0).expand(B, -1), ts, body)
    prompt = torch.randint(0, 50257, (B, n_prompt), generator=g)
    y_in = torch.cat([torch.full((B, 1), 50361), prompt, torch.tensor([50258, 50261, 50359]).expand(B, -1), body], dim=1)
    y_out = torch.cat([y_in[:, 1:], torch.full((B, 1), 50257)], dim=1)
    y_out[:, :n_prompt + 1] = -100  # the prompt (and the step that predicts sot) carries no loss
    return y_in.to(device), y_out.to(device)


def cpu_baseline(model_name: str, S: int, budget_s: float = 30.0):
    """Oracle fwd+bwd (fp32, torch CPU) on ONE synthetic clip of the same architecture, or of a
    smaller one if a forward pass alone would blow the time budget."""
    sys.path.insert(0, str(ROOT))
    from oracle import whisper_oracle as O

    # measured on the GPU box (128 cores / 256 threads): 32-64 torch threads are 2x faster than 128 on this
    # many-small-GEMM workload, so the baseline uses at most 64 (the count actually used is reported)
    cores = min(torch.get_num_threads(), 64)
    torch.set_num_threads(cores)
    name = model_name
    dims = O.DIMS[name]
    params = {k: v.requires_grad_(k != "encoder.positional_embedding") for k, v in O.init_params(dims, seed=0).items()}
    audio, y_in, y_out = O.synthetic_batch(dims, 1, S)
    t0 = time.perf_counter()
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    loss = O.cross_entropy(O.Oracle(dims, params).forward(mel, y_in), y_out, 0.1)
    loss.backward()
    dt = time.perf_counter() - t0
    return {
        "value": round(30.0 / dt, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
        "sample": f"1 synthetic 30 s clip, whisper-{name} fp32, S={S}: log-mel + forward + CE + backward "
                  f"(no optimizer step), {dt:.1f} s on {cores} torch threads ({os.cpu_count()} logical CPUs)",
    }


def cpu_baseline_small(budget_s: float = 40.0):
    """the oracle's forward + backward + AdamW step for whisper-tiny and whisper-base at B = 2, S = 128 on all host
    cores (capped at 64 threads, as cpu_baseline) and on ONE core, while the time budget lasts."""
    sys.path.insert(0, str(ROOT))
    from oracle import whisper_oracle as O

    out, t_start = [], time.perf_counter()
    all_cores = min(os.cpu_count() or 1, 64)
    for name in ("tiny", "base"):
        for cores in (all_cores, 1):
            if time.perf_counter() - t_start > budget_s:
                return out
            torch.set_num_threads(cores)
            dims = O.DIMS[name]
            params = {k: v.requires_grad_(k != "encoder.positional_embedding") for k, v in O.init_params(dims, seed=0).items()}
            opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-5)
            audio, y_in, y_out = O.synthetic_batch(dims, 2, 128)
            t0 = time.perf_counter()
            mel = O.log_mel_spectrogram(audio, dims.n_mels)
            loss = O.cross_entropy(O.Oracle(dims, params).forward(mel, y_in), y_out, 0.1)
            loss.backward()
            opt.step()
            dt = time.perf_counter() - t0
            out.append({"model": name, "cores": cores, "value": round(60.0 / dt, 2), "unit": "audio-s/s", "kind": "port",
                        "sample": f"2 synthetic 30 s clips, fp32, S=128: log-mel + forward + CE + backward + AdamW, {dt:.2f} s"})
    torch.set_num_threads(all_cores)
    return out


class Case:
    """One model configuration resident on the device (model, optimizer, optional DDP wrapper, GPU front end) that can be
    timed at several (clips, S) shapes.  A step is the product loop: `model_utils.train_step` (the reference's
    model/model_utils.py:23-127 — micro-batch fetch, forward, label-smoothed CE, backward, per-micro-batch loss.item(),
    clip folded into the libwft optimizer, optimizer + scheduler step, zero_grad) pulling its micro-batch from an iterator
    that runs log-mel + SpecAugment on clips already resident in HBM."""

    def __init__(self, args, device, rank, local_rank, world, ddp, lora=False, muon=False, sd=0.0, dsa=False, model_name=None,
                 prompt_ts=False):
        import whisper_finetune.runtime as rt
        from whisper_finetune.data.gpu_frontend import GpuFrontend
        from whisper_finetune.model.optimizer import WftAdamW, get_optimizer

        self.args, self.device, self.rank, self.local_rank, self.world, self.ddp = args, device, rank, local_rank, world, ddp
        self.lora, self.muon, self.sd, self.dsa, self.prompt_ts = lora, muon, sd, dsa, prompt_ts
        self.model_name = model_name or args.model
        self.model, self.dims = build_model(self.model_name, device, sd)
        if lora:
            from whisper_finetune.model.lora import apply_lora

            apply_lora(self.model, {"rank": 16, "lora_alpha": 32, "lora_dropout": 0.1})
        if dsa:
            from whisper_finetune.model.model_utils import register_deep_spec_augment_hooks

            register_deep_spec_augment_hooks(self.model, 100, 43)
        self.model.train()
        self.frontend = GpuFrontend(self.dims.n_mels, device, spec_augment=True,
                                    spec_augment_params={"time_mask_param": 100, "freq_mask_param": 43, "time_warp_w": 80, "p": 1.0})
        if muon:
            self.opt = get_optimizer(self.model, {"type": "adamw", "muon": True, "8bit": False,
                                                  "muon_params": {"lr": 2e-5, "momentum": 0.95, "weight_decay": 0.01},
                                                  "params": {"lr": 2e-5, "weight_decay": 0.01, "betas": [0.9, 0.98], "eps": 1e-6}},
                                     is_lora_run=lora)
        else:
            # one wft_mt_sumsq + one wft_mt_adamw launch per step: clip_grad_norm_(1.0) folded into the AdamW pass
            self.opt = WftAdamW([p for p in self.model.parameters() if p.requires_grad], lr=1e-5, betas=(0.9, 0.98), eps=1e-6,
                                weight_decay=0.1)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: 1.0)
        self.net = self.model
        if ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP

            self.net = DDP(self.model, device_ids=[local_rank], output_device=local_rank, broadcast_buffers=False,
                           gradient_as_bucket_view=True, bucket_cap_mb=rt.ddp_bucket_cap_mb(self.model), find_unused_parameters=sd > 0)
        self.t_cfg = {"mixed_precision_training": True, "mp_dtype": "bf16", "accum_grad_steps": 1, "max_grad_norm": 1.0,
                      "label_smoothing": 0.1, "is_lora_run": False,
                      # (A/B hook: WFT_DEFER_LOSS=0 puts the reference's synchronous loss.item() back in front of the optimizer step)
                      "wft_defer_loss_readback": os.environ.get("WFT_DEFER_LOSS", "1") != "0"}

    def mode(self):
        return (("LoRA r=16 alpha=32 p=0.1" if self.lora else "full fine-tune") + (", Muon+AuxAdam" if self.muon else ", AdamW")
                + (f", stochastic depth {self.sd}" if self.sd > 0 else "") + (", deep SpecAugment" if self.dsa else ""))

    def fence(self):
        if self.ddp:
            dist.barrier(device_ids=[self.local_rank])
        torch.cuda.synchronize()

    def step_flops(self, B, S):
        f_fwd, f_att = fwd_flops_per_clip(self.dims, S), attn_core_flops_per_clip(self.dims, S)
        if self.lora:  # frozen base weights: no dW GEMMs
            return (2.0 * (f_fwd - f_att) + 3.0 * f_att + 3.0 * lora_flops_per_clip(self.dims, S, 16)) * B
        return 3.0 * f_fwd * B

    def measure(self, B, S, steps, warmup, roofline=True, hand_rolled_steps=0, ddp_twin=True):
        """-> dict(value, ms_per_step, ...) for `B` clips per GPU and decoder length S; max over ranks of the wall time of
        exactly `steps` optimizer steps between barrier + synchronize."""
        import whisper_finetune.runtime as rt
        from whisper_finetune.engine import kernels as K
        from whisper_finetune.model.model_utils import train_step

        dev, rank = self.device, self.rank
        torch.manual_seed(1234 + rank)  # per-rank host RNG (finetune.py:325)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        audio = torch.randn(B, 480000, device=dev, generator=gen) * 0.1  # resident in HBM
        y_in, y_out = synthetic_tokens(B, S, dev, rank, self.prompt_ts)
        S = y_in.shape[1]  # (prompt_ts: S transcript positions + 65 prompt positions)
        frontend, net, opt, sched = self.frontend, self.net, self.opt, self.sched

        def batches():
            while True:
                yield frontend(audio, training=True), y_in, y_out

        it = batches()
        rt.IS_DISTRIBUTED = self.world > 1 or getattr(self, "force_distributed", False)

        def step():
            return train_step(net, it, opt, sched, self.t_cfg)

        def hand_rolled():  # round 1's loop: same kernels, no per-micro-batch loss.item(), no scheduler
            mel = frontend(audio, training=True)
            loss = net(mel, y_in, targets=y_out, label_smoothing=0.1)
            loss.backward()
            opt.fuse_clip_grad_norm(1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss

        for _ in range(warmup):
            step()
        self.fence()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            loss = step()
            per_step.append(time.perf_counter() - t1)  # (train_step ends with loss.item(): this step's device work is done)
        self.fence()
        dt = time.perf_counter() - t0
        if self.ddp:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        res = {"batch": B, "seq_len": S, "ms_per_step": round(dt / steps * 1e3, 2),
               "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2] * 1e3, 2),
               "value": round(B * self.world * steps * 30.0 / dt, 1), "final_loss": round(float(loss), 4)}
        tf = self.step_flops(B, S) / (dt / steps) / 1e12
        res["step_tflops_per_gpu"] = round(tf, 1)
        res["step_frac_of_bf16_peak"] = round(tf / PEAK_BF16_TFLOPS, 4)
        if self.ddp and ddp_twin:
            # The multi-GPU line describes itself: ranks, bucket size, and what the gradient exchange costs in
            # the step.  Forward + backward only (no optimizer step: under no_sync() every rank would apply its local, un-reduced
            # gradients and the replicas would diverge for the rest of the run — ADVICE r4), once with the all-reduce and once under
            # no_sync(), one untimed warm-up pass each; the difference is the exposed communication + reducer work (the part NOT
            # hidden under the backward).
            import contextlib

            def fwd_bwd(sync: bool):
                with (contextlib.nullcontext() if sync else net.no_sync()):
                    mel = frontend(audio, training=True)
                    net(mel, y_in, targets=y_out, label_smoothing=0.1).backward()
                opt.zero_grad(set_to_none=True)

            twin = {}
            for sync in (True, False):
                fwd_bwd(sync)
                self.fence()
                t0 = time.perf_counter()
                for _ in range(min(3, steps)):
                    fwd_bwd(sync)
                self.fence()
                t = torch.tensor([(time.perf_counter() - t0) / min(3, steps)], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                twin[sync] = t.item() * 1e3
            res["ddp"] = {"rccl_ranks": self.world, "bucket_cap_mb": rt.ddp_bucket_cap_mb(self.model), "gradient_as_bucket_view": True,
                          "launch_mode": "persistent grids; per-tile launches (wft_gemm_args / wft_attn_args launch_mode = 1) for the backward "
                                         "pass of the last micro-batch of a window, beside the gradient exchange (runtime.exchange_launch_mode)",
                          "fwd_bwd_ms_all_reduce": round(twin[True], 2), "fwd_bwd_ms_no_sync": round(twin[False], 2),
                          "exposed_exchange_ms": round(twin[True] - twin[False], 2)}
        if hand_rolled_steps > 0:
            self.fence()
            t0 = time.perf_counter()
            for _ in range(hand_rolled_steps):
                hand_rolled()
            self.fence()
            res["hand_rolled_ms_per_step"] = round((time.perf_counter() - t0) / hand_rolled_steps * 1e3, 2)
        if roofline:
            # one more step with HIP events around every gemm_nt launch (on the launch stream).  EVERY rank runs the step (its
            # gradient all-reduce is a collective); only rank 0 records.
            if rank == 0:
                K.PROFILE_NT = []
            graph_mode = self.t_cfg.get("wft_hip_graph", False)
            self.t_cfg["wft_hip_graph"] = False  # (the HIP events are recorded by Python around each launch: an eager step)
            step()
            self.t_cfg["wft_hip_graph"] = graph_mode
            torch.cuda.synchronize()
            if rank == 0:
                recs, K.PROFILE_NT = K.PROFILE_NT, None
                # the roofline kernel = the NT kernel family with the most time in the step (4: gemm_nt4w_kernel, 256: the ping-pong
                # kernel, 128: the small-tile kernel — whisper-base at 8 clips)
                names = {4: "gemm_nt4w_kernel", 256: "gemm_nt256_kernel", 128: "gemm_nt_kernel"}
                by_v = {}
                for s_, e_, f, v, nb in recs:
                    by_v.setdefault(v, []).append((s_.elapsed_time(e_), f, nb))
                vbig = max(by_v, key=lambda v: sum(t for t, _, _ in by_v[v]))
                big, kname = by_v[vbig], names.get(vbig, str(vbig))
                ms = sum(t for t, _, _ in big)
                flops = sum(f for _, f, _ in big)
                all_ms = sum(s_.elapsed_time(e_) for s_, e_, _, _, _ in recs)
                all_fl = sum(f for _, _, f, _, _ in recs)
                ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                traffic, traffic_note = pmc_traffic_per_launch(B, self.lora, kname) if (kname in ("gemm_nt256_kernel", "gemm_nt4w_kernel") and self.model_name == "large-v3" and not self.prompt_ts) else (None, "no PMC pass committed for this configuration")
                res["roofline"] = {
                    "kernel": kname, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "algorithmic_bytes_per_launch_avg": round(sum(nb for _, _, nb in big) / max(len(big), 1)),
                    "launches": len(big), "avg_launch_us": round(ms * 1e3 / max(len(big), 1), 2),
                    "flops_per_launch_avg": round(flops / max(len(big), 1)),
                    "all_nt_gemm_launches": len(recs),
                    "all_nt_gemm_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 1) if all_ms > 0 else 0.0,
                }
        rt.IS_DISTRIBUTED = False
        del audio
        return res

    def release(self):
        self.net = self.model = self.opt = self.sched = None
        import gc

        gc.collect()
        torch.cuda.empty_cache()


def ddp_mode_1gpu(case, B, S, plain_ms):
    """the WORLD_SIZE = 8 launch configuration as a measured mode on ONE GPU.  The headline model is wrapped in
    DDP over a 1-rank RCCL group (what WFT_BENCH_FORCE_DDP=1 does) and re-timed with per-tile NT / dK/dV launches (what
    engine/lib.py selects when WORLD_SIZE > 1) and with persistent ones, each (a) with the 1-rank all-reduce and (b) with a DDP
    communication hook that launches, per 64 MB gradient bucket and on a side stream, a kernel that HOLDS `thief_cus` CUs and
    moves the bytes an 8-rank ring moves for that bucket (2 * 7/8 of it, read and written) paced at `thief_gbps` — RCCL's
    footprint on this rank (10.8 GB per step).  "persistent vs per-tile under a CU thief" is decided by these four
    numbers, not by a comment."""
    import ctypes

    from torch.nn.parallel import DistributedDataParallel as DDP

    from whisper_finetune.engine import lib as L

    lib = L.load()
    dev = case.device
    thief_cus, thief_gbps = 24, 250.0
    side_lib = ctypes.CDLL(str(ROOT / "whisper-finetune_amd" / "libwft_bench.so"))
    side_lib.wft_bench_cu_thief.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_void_p]
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    # a DDP job from its first step: no gradient homes left over from the plain run (they would keep the plain run's gradient
    # tensors alive beside the buckets), peak-memory statistics from here on
    for prm in case.model.parameters():
        prm.__dict__.pop("_wft_grad_home", None)
        prm.grad = None
    case.opt.__dict__["_wft_steps"] = 0
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    scratch = torch.empty(2, 256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    moved = [0]

    def thief_hook(state, bucket):
        buf = bucket.buffer()
        nbytes = int(buf.numel() * buf.element_size() * 2 * 7 / 8)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rc = side_lib.wft_bench_cu_thief(scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[0].numel(), nbytes, thief_cus,
                                             thief_gbps, ctypes.c_void_p(side.cuda_stream))
            assert rc == 0, rc
            moved[0] += nbytes
            fut = torch.futures.Future(devices=[dev])
            fut.set_result(buf)  # (a CUDA-aware future: consumers wait for the side stream's work up to here)
        return fut

    out = {"batch": B, "seq_len": S, "plain_ms_per_step": plain_ms, "thief_cus": thief_cus, "thief_gbps": thief_gbps,
           "what": "headline workload under DDP (1-rank RCCL group, gradient_as_bucket_view, 64 MB buckets); per_tile = "
                   "launch_mode = 1 in every wft_gemm_args / wft_attn_args (kernels.LAUNCH_OVERRIDE); product = train_step's multi-process mode: "
                   "persistent grids, per-tile launches only for the backward pass beside the gradient exchange; thief = a "
                   "side-stream kernel per gradient bucket holding thief_cus CUs and copying 2*7/8 of the bucket at thief_gbps"}
    from whisper_finetune.engine import kernels as K_

    saved = (case.net, case.ddp)
    try:
        for thief in (False, True):
            case.net = DDP(case.model, device_ids=[case.local_rank], output_device=case.local_rank, broadcast_buffers=False,
                           gradient_as_bucket_view=True, bucket_cap_mb=64)
            case.ddp = True
            if thief:
                case.net.register_comm_hook(None, thief_hook)
            # product = what model_utils.train_step does in a multi-process job: persistent grids, per-tile launches only for the
            # backward pass that runs beside the gradient exchange (runtime.exchange_launch_mode)
            for mode, force in (("persistent", 0), ("per_tile", 1), ("product", None)):
                K_.LAUNCH_OVERRIDE[0] = force  # None: the per-call argument train_step's launch mode sets decides
                case.force_distributed = mode == "product"
                moved[0] = 0
                r = case.measure(B, S, 3, 2 if (not thief and mode == "persistent") else 1, roofline=False,
                                 ddp_twin=(not thief and mode == "persistent"))
                key = f"ddp_{mode}" + ("_thief" if thief else "")
                out[key + "_ms_per_step"] = r["ms_per_step"]
                if thief:
                    out["thief_gb_per_step"] = round(moved[0] / 4 / 1e9, 2)  # (3 timed + 1 warm-up step)
                if not thief and mode == "persistent":
                    # peak over a DDP job's first steps (torch DDP allocates its rebuilt buckets beside the first iteration's at the
                    # start of its second iteration: +6 GB of its own) ...
                    out["hbm_peak_gib_under_ddp"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
                    torch.cuda.reset_peak_memory_stats()
                if not thief and mode == "per_tile":
                    # ... and in the steady state behind it (buckets rebuilt, gradient homes = the bucket views)
                    out["hbm_peak_gib_under_ddp_steady_state"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
                if "ddp" in r:
                    out["exposed_exchange_ms_1rank"] = r["ddp"]["exposed_exchange_ms"]
            case.net = None
        out["ddp_persistent_overhead_pct"] = round((out["ddp_persistent_ms_per_step"] / plain_ms - 1.0) * 100.0, 2)  # the default launch mode
        out["ddp_per_tile_overhead_pct"] = round((out["ddp_per_tile_ms_per_step"] / plain_ms - 1.0) * 100.0, 2)
        out["ddp_product_overhead_pct"] = round((out["ddp_product_ms_per_step"] / plain_ms - 1.0) * 100.0, 2)  # what a multi-GPU job runs
        out["per_tile_vs_persistent_under_thief_pct"] = round((out["ddp_per_tile_thief_ms_per_step"] / out["ddp_persistent_thief_ms_per_step"] - 1.0) * 100.0, 2)
    finally:
        K_.LAUNCH_OVERRIDE[0] = None
        case.force_distributed = False
        case.net, case.ddp = saved
        del scratch
        if own_pg:
            dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=96, help="clips per GPU per step")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--lora", action="store_true", help="BASELINE configs[2]: LoRA r=16 alpha=32 dropout 0.1 on every Linear")
    ap.add_argument("--muon", action="store_true", help="Muon + auxiliary Adam param groups (config_large_v3_best_muon.yaml)")
    ap.add_argument("--stochastic-depth", type=float, default=0.0)
    ap.add_argument("--deep-spec-augment", action="store_true")
    ap.add_argument("--hip-graph", action="store_true", help="training.wft_hip_graph: the micro-batch as one captured HIP graph (small models)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline line: skip the reference-YAML shapes (B=32 at S=128 / S=448), the hand-rolled-loop "
                         "cross-check and the LoRA+Muon (configs[2]) sub-bench")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = os.environ.get("WFT_BENCH_FORCE_DDP") == "1"  # exercise the DDP wrapper on one GPU (1-rank RCCL group)
    ddp = world > 1 or force_ddp
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", rank=0, world_size=1)
        else:
            dist.init_process_group(backend="nccl")  # RCCL over xGMI

    from whisper_finetune.engine import lib as L

    L.load()
    B, S = args.batch, args.seq
    headline_default = args.model == "large-v3" and not (args.lora or args.muon or args.stochastic_depth > 0 or args.deep_spec_augment)
    # the extra lines are single-GPU measurements (the driver's N > 1 runs time the headline only: weak scaling of ONE workload)
    extras = headline_default and not args.no_extras and world == 1
    case = Case(args, device, rank, local_rank, world, ddp, lora=args.lora, muon=args.muon, sd=args.stochastic_depth,
                dsa=args.deep_spec_augment)
    if args.hip_graph:
        case.t_cfg["wft_hip_graph"] = True
    head = case.measure(B, S, args.steps, args.warmup, roofline=not args.no_roofline, hand_rolled_steps=3 if extras else 0)
    hbm_peak = round(torch.cuda.max_memory_allocated() / 2**30, 1)
    other = []
    lora_line = None
    if extras:
        # the shapes a user of the reference's YAML runs (configs/config_large_v3_best_muon.yaml:56 batch_size 32): the primary
        # S = 128 and the max-context stress S = 448; same model / optimizer state, 2 + 4 steps each
        for b2, s2 in ((32, 128), (32, 448)):
            other.append(case.measure(b2, s2, 4, 2, roofline=False))
        try:
            ddp_block = ddp_mode_1gpu(case, B, S, head["ms_per_step"])
        except Exception as exc:  # informative; never lose the headline over it
            ddp_block = {"failed": repr(exc)}
        mode = case.mode()
        dims = case.dims
        case.release()
        # BASELINE configs[2]: large-v3 LoRA r16 + Muon/AuxAdam + stochastic depth 0.1 + deep SpecAugment at the YAML's B = 32
        c2 = Case(args, device, rank, local_rank, world, ddp, lora=True, muon=True, sd=0.1, dsa=True)
        lora_line = c2.measure(32, 128, 4, 2, roofline=not args.no_roofline)
        lora_line["workload"] = f"whisper-large-v3 {c2.mode()}, 32 clips per GPU per step, S=128 (BASELINE configs[2] shape)"
        c2.release()
        # BASELINE configs[1]: whisper-base full fine-tune, 8 clips (host-bound: 810 launches of 13 ms of kernels per step)
        # Launch-bound eagerly; `training.wft_hip_graph: true` runs each micro-batch (forward + fused loss + backward) as one captured
        # HIP graph (engine/graph.py; bit-identical to the eager path) — the line is the graph mode, the eager step time is beside it
        c1 = Case(args, device, rank, local_rank, world, ddp, model_name="base")
        eager = c1.measure(8, 128, 20, 5, roofline=False)
        c1.t_cfg["wft_hip_graph"] = True
        base_line = c1.measure(8, 128, 30, 8, roofline=not args.no_roofline)
        base_line["eager_ms_per_step"], base_line["eager_ms_per_step_median"] = eager["ms_per_step"], eager["ms_per_step_median"]
        from whisper_finetune.engine import graph as _G
        gm = _G.graphed_for(c1.model)
        base_line["hip_graph"] = {"micro_batch_graphs": len(gm[1].graphs) if gm else 0, "disabled": gm[1].disabled if gm else "not built"}
        base_line["workload"] = (f"whisper-base {c1.mode()}, 8 clips per GPU per step, S=128 (BASELINE configs[1]), training.wft_hip_graph: "
                                 "forward + loss + backward of the micro-batch as one HIP graph; front end, clip, optimizer, scheduler eager")
        c1.release()
        # BASELINE configs[4]'s per-GPU workload: large-v3-turbo (4-layer decoder), LoRA r16, prompt + timestamp targets, B = 64
        c4 = Case(args, device, rank, local_rank, world, ddp, lora=True, model_name="large-v3-turbo", prompt_ts=True)
        turbo_line = c4.measure(64, 128, 4, 2, roofline=not args.no_roofline)
        turbo_line["workload"] = (f"whisper-large-v3-turbo {c4.mode()}, 64 clips per GPU per step, 64-token prompt (-100 targets) + 128 "
                                  "transcript positions with a timestamp token every 16th (BASELINE configs[4] per-GPU shape)")
        c4.release()
        # the entrypoint's real loop (DataLoader workers -> GpuMelLoader -> train_step) next to the same batches from HBM
        try:
            sys.path.insert(0, str(ROOT / "tools"))
            from e2e_entrypoint import measure as e2e_measure

            entry_line = e2e_measure("large-v3", 32, steps=8, warmup=3, device=device)
        except Exception as exc:  # informative; never lose the headline over it
            entry_line = {"failed": repr(exc)}
        import gc

        gc.collect()
        torch.cuda.empty_cache()
    else:
        mode, dims = case.mode(), case.dims
        base_line = turbo_line = entry_line = ddp_block = None

    if rank == 0:
        out = {
            "metric": "audio-seconds/sec training throughput, whisper-large-v3 bf16",
            "value": head["value"], "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "ms_per_step_median": head["ms_per_step_median"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": f"whisper-{args.model} {mode}, bf16 MFMA / fp32 master weights, {B} synthetic 30 s clips per GPU "
                            f"per step, decoder S={S}, log-mel + SpecAugment on GPU, label smoothing 0.1, clip 1.0, "
                            f"local accumulation 1 (global window = {world}); timed loop = model_utils.train_step",
                "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
            },
            "step_tflops_per_gpu": head["step_tflops_per_gpu"],
            "step_frac_of_bf16_peak": head["step_frac_of_bf16_peak"],
            "final_loss": head["final_loss"],
            "hbm_peak_gib": hbm_peak,
            "roofline": head.get("roofline"),
        }
        if "ddp" in head:
            out["ddp"] = head["ddp"]  # multi-GPU (or WFT_BENCH_FORCE_DDP=1): ranks, bucket size, the no_sync twin, exposed exchange
        if "hand_rolled_ms_per_step" in head:
            out["hand_rolled_ms_per_step"] = head["hand_rolled_ms_per_step"]  # round 1's loop, 3 steps: cross-check of the product loop
        if ddp_block is not None:
            out["ddp_mode_1gpu"] = ddp_block
        if other:
            out["reference_yaml_shapes"] = other
        if lora_line is not None:
            out["configs2_lora_muon"] = lora_line
        if base_line is not None:
            out["configs1_base"] = base_line
        if turbo_line is not None:
            out["configs4_turbo_lora"] = turbo_line
        if entry_line is not None:
            out["entrypoint_loop"] = entry_line
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.model if args.model in ("tiny", "base", "small", "large-v3", "large-v3-turbo") else "large-v3", S)
            except Exception as exc:  # the baseline is informative; never lose the GPU number over it
                out["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {exc!r}"}
            if extras:
                try:
                    out["cpu_baseline_small_models"] = cpu_baseline_small()
                except Exception as exc:
                    out["cpu_baseline_small_models"] = [{"failed": repr(exc)}]
        else:
            out["cpu_baseline"] = None
    if ddp:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes version / path banners through C stdio (flushed only at exit): push them out first so that the JSON
        # line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
