"""Worker functions of the multi-process GPU tests (spawned by tests/test_dist_gpu.py; not collected by pytest)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _setup(rank, world, port, backend):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      WFT_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (str(ROOT), str(ROOT / "whisper-finetune_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def sharded_muon_worker(rank, world, port, out):
    """Two processes on ONE GPU over gloo: the rank-sharded Muon step (owner-computed Newton-Schulz + all-gathered bf16
    updates) must give every rank exactly the parameters of the unsharded step."""
    _setup(rank, world, port, "gloo")
    import torch
    import torch.distributed as dist
    from whisper_finetune.model import optimizer as wopt

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")

    class Stack(torch.nn.Module):
        def __init__(self, n, d):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(d, d) for _ in range(n)])

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder, self.decoder = Stack(3, 256), Stack(2, 256)   # 5 square matrices: uneven deal over 2 ranks
            self.wide = Stack(1, 128)
            self.decoder.blocks.append(torch.nn.Linear(384, 128))       # one wide matrix: a bucket smaller than the world
            self.embedding = torch.nn.Embedding(16, 256)

    conf = {"type": "adamw", "muon": True, "8bit": False, "muon_ndim_threshold": 2,
            "muon_params": {"lr": 1e-2, "momentum": 0.95, "weight_decay": 0.01},
            "params": {"lr": 1e-3, "weight_decay": 0.01, "betas": [0.9, 0.98], "eps": 1e-6}}
    torch.manual_seed(0)
    a = Fake().to(dev)
    b = Fake().to(dev)
    b.load_state_dict(a.state_dict())
    opt_a = wopt.get_optimizer(a, conf)          # sharded over the 2-rank group
    real = wopt._muon_shard
    wopt._muon_shard = lambda: None
    opt_b = wopt.get_optimizer(b, conf)          # the single-process form
    for step in range(3):
        g = torch.Generator(device="cpu").manual_seed(100 + step)  # identical (all-reduced) gradients on every rank
        for pa, pb in zip(a.parameters(), b.parameters()):
            gr = torch.randn(pa.shape, generator=g).to(dev)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        wopt._muon_shard = real
        opt_a.fuse_clip_grad_norm(1.0)
        opt_a.step()
        wopt._muon_shard = lambda: None
        opt_b.fuse_clip_grad_norm(1.0)
        opt_b.step()
    worst = max((pa.detach() - pb.detach()).abs().max().item() for pa, pb in zip(a.parameters(), b.parameters()))
    n_state = sum(1 for p in a.parameters() if "momentum_buffer" in opt_a.state.get(p, {}))
    n_muon = sum(len(g["params"]) for g in opt_a.param_groups if g["use_muon"])
    flat = torch.cat([p.detach().flatten() for p in a.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, worst, n_state, n_muon, same))


def ddp_engine_worker(rank, world, port, out):
    """The ENGINE under torch DDP on a 1-rank RCCL group: stochastic depth (different unused parameters per step),
    find_unused_parameters=True, gradient_as_bucket_view=True, local accumulation 2 (no_sync on the first micro-batch),
    the libwft optimizer reading the bucket views, per-tile GEMM launches — against the same steps without DDP."""
    _setup(rank, world, port, "nccl")
    os.environ["WFT_NT256_PERSISTENT"] = "0"  # what engine/lib.py selects when WORLD_SIZE > 1 (RCCL kernels hold CUs; measured: bench.py ddp_mode_1gpu)
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import whisper_finetune.runtime as rt
    from oracle import whisper_oracle as O
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.model import model_utils
    from whisper_finetune.model.model_utils import CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder
    from whisper_finetune.model.optimizer import WftAdamW

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    dims = O.DIMS["tiny"]
    params = O.init_params(dims, seed=2)

    def build():
        m = Whisper(ModelDimensions(**vars(dims)))
        m.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head,
                                                       dims.n_audio_layer, 0.3)
        m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                      dims.n_text_layer, 0.3)
        m.load_state_dict(params)
        return m.to(dev)

    audio, y_in, y_out = O.synthetic_batch(dims, 2, 16)
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 2, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1}

    def batches():
        while True:
            yield mel, y_in, y_out

    def run(wrap):
        m = build()
        model = DDP(m, device_ids=[0], output_device=0, find_unused_parameters=True, broadcast_buffers=False,
                    gradient_as_bucket_view=True, bucket_cap_mb=64) if wrap else m
        opt = WftAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
        entries = {"n": 0}
        if wrap:
            orig = model.no_sync

            def counting():
                entries["n"] += 1
                return orig()

            model.no_sync = counting
        rt.IS_DISTRIBUTED = wrap  # train_step enters no_sync() on all but the last micro-batch only in a distributed job
        torch.manual_seed(5)      # same stochastic-depth draws in both runs
        losses = [model_utils.train_step(model, batches(), opt, sched, dict(t_cfg)) for _ in range(3)]
        rt.IS_DISTRIBUTED = False
        unused = [n for n, p in m.named_parameters() if p.grad is not None and p.grad.abs().sum() == 0]
        return losses, {n: p.detach().clone() for n, p in m.named_parameters()}, entries["n"], unused

    l_plain, p_plain, _, _ = run(False)
    l_ddp, p_ddp, n_nosync, _ = run(True)
    worst = max((p_plain[n] - p_ddp[n]).abs().max().item() for n in p_plain)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    out.put((rank, l_plain, l_ddp, worst, n_nosync))


def ddp_engine_two_rank_worker(rank, world, port, out):
    """TWO processes on one GPU over gloo (RCCL refuses two ranks per device; gloo reduces HIP tensors through the host): the
    engine under a real 2-rank DDP reducer — per-rank seeds, so stochastic depth drops DIFFERENT blocks on the two ranks
    (find_unused_parameters=True: the used-parameter bitmap exchange), local accumulation 2 with no_sync()
    on the first micro-batch, gradient_as_bucket_view.  Checked against a hand-made reduction: every rank replays its own
    micro-batches on a plain copy of the model, the gradients are averaged with explicit all_reduce calls (a parameter unused
    on BOTH ranks keeps grad None, as DDP leaves it), and the same optimizer step must give the same parameters."""
    _setup(rank, world, port, "gloo")
    os.environ["WFT_NT256_PERSISTENT"] = "0"
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import whisper_finetune.runtime as rt
    from oracle import whisper_oracle as O
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.model import model_utils
    from whisper_finetune.model.model_utils import CheckpointedStochasticAudioEncoder, CheckpointedStochasticTextDecoder
    from whisper_finetune.model.optimizer import WftAdamW

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    dims = O.DIMS["tiny"]
    params = O.init_params(dims, seed=2)

    def build():
        m = Whisper(ModelDimensions(**vars(dims)))
        m.encoder = CheckpointedStochasticAudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head,
                                                       dims.n_audio_layer, 0.4)
        m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                                      dims.n_text_layer, 0.4)
        m.load_state_dict(params)
        return m.to(dev).train()

    audio, y_in, y_out = O.synthetic_batch(dims, 2, 16, seed=50 + rank)  # every rank its own shard
    mel = O.log_mel_spectrogram(audio, dims.n_mels).to(dev)
    y_in, y_out = y_in.to(dev), y_out.to(dev)
    kw = dict(lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    accum = 2
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": accum, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1}

    # ---- DDP run
    m = build()
    ddp = DDP(m, device_ids=[0], output_device=0, find_unused_parameters=True, broadcast_buffers=False, gradient_as_bucket_view=True,
              bucket_cap_mb=64)
    opt = WftAdamW(m.parameters(), **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    rt.IS_DISTRIBUTED, rt.WORLD_SIZE, rt.RANK = True, world, rank

    def batches():
        while True:
            yield mel, y_in, y_out

    torch.manual_seed(100 + rank)  # per-rank host RNG (scripts/finetune.py:325): different blocks dropped per rank
    loss = model_utils.train_step(ddp, batches(), opt, sched, dict(t_cfg))
    rt.IS_DISTRIBUTED = False

    # ---- hand-made reduction on a plain copy
    ref = build()
    ropt = WftAdamW(ref.parameters(), **kw)
    torch.manual_seed(100 + rank)
    rloss = 0.0
    for _ in range(accum):
        l = ref(mel, y_in, targets=y_out, label_smoothing=0.1) / accum
        l.backward()
        rloss += l.item()
    n_local_unused = 0
    for p in ref.parameters():
        used = torch.tensor([0.0 if p.grad is None else 1.0])
        n_local_unused += int(p.grad is None)
        dist.all_reduce(used)
        g = torch.zeros_like(p) if p.grad is None else p.grad.clone()
        g_host = g.cpu()
        dist.all_reduce(g_host)
        p.grad = (g_host / world).to(dev) if used.item() > 0 else None
    ropt.fuse_clip_grad_norm(1.0)
    ropt.step()
    worst = max((a.detach() - b.detach()).abs().max().item() for a, b in zip(m.parameters(), ref.parameters()))
    flat = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, loss, rloss, worst, same, n_local_unused))


def ddp_grad_homes_worker(rank, world, port, out):
    """whisper-base (d = 512: its weight-gradient GEMMs run on the 256x256 paths at 12 clips) under DDP on a 1-rank RCCL group with
    gradient_as_bucket_view and local accumulation 2: from the third optimizer step on (DDP rebuilds its buckets after the first
    iteration, the optimizer notes the new views at the second step) every large weight gradient is written by the GEMM's reduce
    kernel straight into its bucket view — no reducer copy — and the parameters equal the run without DDP bit for bit."""
    _setup(rank, world, port, "nccl")
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import whisper_finetune.runtime as rt
    from oracle import whisper_oracle as O
    from whisper_finetune.engine import kernels as K
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.model import model_utils
    from whisper_finetune.model.optimizer import WftAdamW

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    dims = O.DIMS["base"]
    params = O.init_params(dims, seed=3)
    audio, y_in, y_out = O.synthetic_batch(dims, 12, 16)
    mel = K.logmel(audio.to(dev), O.mel_filters(dims.n_mels).to(dev))
    y_in, y_out = y_in.to(dev), y_out.to(dev)
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 2, "max_grad_norm": 1.0, "mp_dtype": "bf16", "label_smoothing": 0.1}

    def batches():
        while True:
            yield mel, y_in, y_out

    seg_calls = {"n": 0, "acc": 0}
    real_tn = K.gemm_tn

    def counting_tn(*a, **kw):
        r = real_tn(*a, **kw)
        if kw.get("seg_out") is not None and r is not None:
            seg_calls["n"] += 1
            seg_calls["acc"] += int(bool(kw.get("accumulate")))
        return r

    def run(wrap):
        m = Whisper(ModelDimensions(**vars(dims)))
        m.load_state_dict(params)
        m.to(dev)
        model = DDP(m, device_ids=[0], output_device=0, broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=64) if wrap else m
        opt = WftAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
        rt.IS_DISTRIBUTED = wrap
        K.gemm_tn = counting_tn
        import whisper_finetune.engine.ops as ops
        alias = []
        try:
            for step in range(4):
                seg_calls["n"] = seg_calls["acc"] = 0
                model_utils.train_step(model, batches(), opt, sched, dict(t_cfg))
                big = [p for n, p in m.named_parameters() if p.dim() == 2 and p.shape[0] % 256 == 0 and p.shape[1] % 256 == 0
                       and "embedding" not in n]
                homes = [p.__dict__.get("_wft_grad_home") for p in big]
                # DDP's bucket views share a handful of storages (290 MB of fp32 gradients in 64 MB buckets); without DDP every
                # Linear group's gradients are slices of that group's own [N, K] product
                alias.append((seg_calls["n"], seg_calls["acc"],
                              len({h.untyped_storage().data_ptr() for h in homes if h is not None}), len(big)))
        finally:
            K.gemm_tn = real_tn
            rt.IS_DISTRIBUTED = False
        return {n: p.detach().clone() for n, p in m.named_parameters()}, alias

    p_plain, a_plain = run(False)
    p_ddp, a_ddp = run(True)
    worst = max((p_plain[n] - p_ddp[n]).abs().max().item() for n in p_plain)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    out.put((rank, worst, a_plain, a_ddp))
