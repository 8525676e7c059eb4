"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Covers what the reference
does around the hot path: runtime.setup_distributed from torchrun-style env,
DistributedSampler sharding (finetune.py:620-627), DDP(find_unused_parameters=True) + train_step with no_sync()
accumulation over a model whose layers are single autograd nodes returning several gradients at once (the shape of
engine/ops.LinearFn) and whose middle block is skipped in different micro-batches on different ranks (stochastic depth),
rank-local loss, identical parameters on every rank after the step, barrier and cleanup."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FusedLinearFn(torch.autograd.Function):
    """CPU stand-in for engine/ops.LinearFn: ONE autograd node that returns the input gradient and BOTH parameter gradients
    at once (the DDP reducer then sees several gradient-ready hooks fire from a single node, in this order)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        return dy @ w, dy.t() @ x, dy.sum(0)


class _Toy(torch.nn.Module):
    """a -> [c: a residual block that stochastic depth may skip] -> b, every layer through _FusedLinearFn.  `skip_c` is
    set per micro-batch by the test (different ranks skip in different micro-batches: the per-rank unused-parameter case of
    the reference's DDP(find_unused_parameters=True), scripts/finetune.py:694-705)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(5, 8)
        self.c = torch.nn.Linear(8, 8)
        self.b = torch.nn.Linear(8, 7)
        self.skip_c = False

    def forward(self, x, y_in):
        h = torch.tanh(_FusedLinearFn.apply(x, self.a.weight, self.a.bias))
        if not self.skip_c:
            h = h + torch.tanh(_FusedLinearFn.apply(h, self.c.weight, self.c.bias))
        h = _FusedLinearFn.apply(h, self.b.weight, self.b.bias)  # [B, 7]
        return h.unsqueeze(1).expand(-1, y_in.shape[1], -1)


class _DS(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(0)
        self.x = torch.randn(n, 5, generator=g)
        self.y = torch.randint(0, 7, (n, 3), generator=g)

    def __len__(self): return len(self.x)
    def __getitem__(self, i): return self.x[i], torch.zeros(3, dtype=torch.long), self.y[i], i


def _collate(items):
    x, yi, yo, idx = zip(*items)
    _collate.seen.extend(idx)
    return torch.stack(x), torch.stack(yi), torch.stack(yo)


_collate.seen = []


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), WFT_DIST_BACKEND="gloo")
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root / "whisper-finetune_amd"))
    import whisper_finetune.runtime as rt
    from whisper_finetune import utils
    from whisper_finetune.model import model_utils
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.utils.data import DataLoader, DistributedSampler

    torch.set_num_threads(1)
    device = rt.setup_distributed()
    assert rt.IS_DISTRIBUTED and rt.WORLD_SIZE == world and rt.RANK == rank and rt.IS_MAIN == (rank == 0)
    assert device.type == "cpu"
    utils.set_seed(100 + rt.RANK)  # per-rank RNG (finetune.py:325)
    torch.manual_seed(0)
    model = _Toy()
    ref = _Toy(); ref.load_state_dict(model.state_dict())
    ddp = DDP(model, broadcast_buffers=False, gradient_as_bucket_view=True, find_unused_parameters=True)
    assert rt.unwrap_model(ddp) is model

    ds = _DS(16)
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=7, drop_last=True)
    loader = DataLoader(ds, batch_size=2, sampler=sampler, collate_fn=_collate, drop_last=True)
    local_accum = utils.resolve_local_accum_grad_steps(4, world)  # global window 4 -> 2 per rank
    t_cfg = {"mixed_precision_training": False, "accum_grad_steps": local_accum, "max_grad_norm": 1e9, "mp_dtype": "bf16",
             "label_smoothing": 0.1}
    opt = torch.optim.SGD(ddp.parameters(), lr=0.5)

    class Sch:
        n = 0
        def step(self): self.n += 1

    entries = {"n": 0}
    orig = ddp.no_sync

    def counting():
        entries["n"] += 1
        return orig()

    ddp.no_sync = counting

    def skips(r, mb):  # which micro-batches of rank r drop block c (rank 0: the first, rank 1: the second)
        return (r + mb) % 2 == 0

    def batches():
        for mb, batch in enumerate(model_utils.infinite_iter(loader)):
            model.skip_c = skips(rank, mb)
            yield batch

    loss = model_utils.train_step(ddp, batches(), opt, Sch(), t_cfg)
    assert entries["n"] == local_accum - 1

    # expected indices of this rank: r::world of the seeded permutation (epoch 0)
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(7 + 0)).tolist()
    mine = perm[rank::world][: 2 * local_accum]
    assert _collate.seen[: 2 * local_accum] == mine, (_collate.seen, mine)

    # reference update: average over ranks of (sum over micro-batches of mean-CE/accum) gradients
    losses = []
    grads = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        idx = perm[r::world][: 2 * local_accum]
        for mb in range(local_accum):
            ii = idx[2 * mb: 2 * mb + 2]
            x = ds.x[ii]; y = ds.y[ii]
            ref.skip_c = skips(r, mb)
            l = torch.nn.functional.cross_entropy(ref(x, y).transpose(1, 2), y, label_smoothing=0.1) / local_accum
            gs = torch.autograd.grad(l, list(ref.parameters()), allow_unused=True)
            for g, gi in zip(grads, gs):
                if gi is not None:
                    g += gi / world
            if r == rank:
                losses.append(l.item())
    with torch.no_grad():
        for p, g in zip(ref.parameters(), grads):
            p -= 0.5 * g
    assert abs(loss - sum(losses)) < 1e-6  # rank-local loss, not all-reduced
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p, q, atol=1e-6, rtol=1e-5)
    # every rank holds identical parameters
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    rt.barrier()
    rt.cleanup()
    out.put((rank, loss))


@pytest.mark.timeout(180)
def test_two_process_gloo_train_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = dict(q.get(timeout=5) for _ in range(2))
    assert set(res) == {0, 1} and res[0] != res[1]  # different shards -> different local losses
