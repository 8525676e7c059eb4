"""8-bit AdamW (reference: model/optimizer.py:241-256, `bnb.optim.AdamW8bit` when `optimizer.8bit: True`).

PARITY UNPINNED: bitsandbytes is not vendored in the reference and not installed here, and the reference's tests hold no vector for
it.  The oracle (oracle/adam8bit_oracle.py) restates the published block-wise dynamic-quantisation scheme; these tests pin what ANY
implementation of that scheme must satisfy (CPU), and the HIP kernel against the restatement (GPU)."""
import numpy as np
import pytest
import torch

from oracle import adam8bit_oracle as A
from whisper_finetune.model import optimizer as wopt


def test_dynamic_maps_are_monotone_256_entry_tables_with_the_published_structure():
    q1, q2 = A.create_dynamic_map(True), A.create_dynamic_map(False)
    for q in (q1, q2):
        assert q.shape == (256,) and q.dtype == np.float32 and np.all(np.diff(q) > 0)
        assert q[-1] == 1.0 and 0.0 in q
    assert np.array_equal(np.sort(-q1[q1 < 0]), q1[(q1 > 0) & (q1 < 1)])  # the negative half mirrors the positive one (1.0 has no twin)
    assert q1[0] < -0.99 and q1[q1 > 0][0] == pytest.approx(5.5e-7) and (q1 < 0).sum() == 127 and (q1 > 0).sum() == 128
    assert q2[0] == 0.0 and (q2 > 0).sum() == 255 and q2[1] == pytest.approx(3.25e-7)
    # decade i holds 2^i (signed) / 2^(i+1) (unsigned) values inside (0.1, 1) * 10^(i-6)
    for i in range(7):
        lo, hi = 0.1 * 10.0 ** (i - 6), 10.0 ** (i - 6)
        assert ((q1 > lo) & (q1 < hi)).sum() == 2 ** i
        assert ((q2 > lo) & (q2 < hi)).sum() == 2 ** (i + 1)
    # the product's own copy of the map (uploaded to the GPU) is this table, bit for bit
    assert np.array_equal(wopt.create_dynamic_map(True).numpy(), q1) and np.array_equal(wopt.create_dynamic_map(False).numpy(), q2)


def test_blockwise_round_trip_error_is_bounded_by_the_local_map_spacing():
    rng = np.random.default_rng(0)
    q1, q2 = A.create_dynamic_map(True), A.create_dynamic_map(False)
    x = (rng.standard_normal(3 * A.BLOCK + 77) * np.exp(rng.uniform(-6, 2, 3 * A.BLOCK + 77))).astype(np.float32)
    for data, q in ((x, q1), (x * x, q2)):
        codes, absmax = A.quantize_blockwise(data, q)
        back = A.dequantize_blockwise(codes, absmax, q)
        assert codes.dtype == np.uint8 and absmax.shape == (4,)
        for b in range(4):
            seg, rec = data[b * A.BLOCK:(b + 1) * A.BLOCK], back[b * A.BLOCK:(b + 1) * A.BLOCK]
            assert absmax[b] == np.abs(seg).max()
            # nearest entry: the error is at most half the gap between the two neighbouring map entries (times absmax)
            n = seg / absmax[b]
            idx = codes[b * A.BLOCK:(b + 1) * A.BLOCK].astype(int)
            gap_lo = q[idx] - q[np.maximum(idx - 1, 0)]
            gap_hi = q[np.minimum(idx + 1, 255)] - q[idx]
            half = 0.5 * np.maximum(gap_lo, gap_hi) * absmax[b]
            assert np.all(np.abs(seg - rec) <= half * (1 + 1e-5) + 1e-12)
            # the block's largest element decodes to absmax itself (code of 1.0) or, negative in the signed map, to the entry next to
            # it; relative error of elements above 10 % of absmax < 0.8 % of absmax (64 / 128 steps per decade)
            assert absmax[b] >= np.abs(rec).max() >= 0.992 * absmax[b]
            big = np.abs(n) > 0.1
            assert np.all(np.abs(rec[big] - seg[big]) <= 0.008 * absmax[b])
        # idempotent: quantising the reconstruction returns the same codes (for blocks whose absmax survived the round trip exactly:
        # a negative extreme in the signed map decodes to -0.993 absmax)
        c2, a2 = A.quantize_blockwise(back, q)
        assert np.all(a2 <= absmax) and np.all(a2 >= 0.992 * absmax)
        for b in range(4):
            if a2[b] == absmax[b]:
                assert np.array_equal(c2[b * A.BLOCK:(b + 1) * A.BLOCK], codes[b * A.BLOCK:(b + 1) * A.BLOCK])


def test_8bit_update_stays_within_one_quantisation_step_of_the_fp32_update():
    """Twenty AdamW steps on a noisy quadratic: the 8-bit-state trajectory tracks the fp32-state one — every parameter update differs
    from the fp32 update by less than the update computed from moments perturbed by one map step, and the loss goes down alike."""
    rng = np.random.default_rng(1)
    n = 3 * A.BLOCK + 500
    target = rng.standard_normal(n).astype(np.float32)
    p8 = np.zeros(n, np.float32)
    p32, m, v = p8.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    st = A.Adam8bitState(n)
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.1)
    for step in range(1, 21):
        noise = rng.standard_normal(n).astype(np.float32) * 0.1
        g8, g32 = (p8 - target) + noise, (p32 - target) + noise
        p8 = A.adamw8bit_step(p8, g8, st, **hp)
        p32, m, v = A.adamw32_step(p32, g32, m, v, step, **hp)
        assert st.state1.dtype == np.uint8 and st.absmax1.shape == (4,)
    # both descend
    l0, l8, l32 = np.mean(target ** 2), np.mean((p8 - target) ** 2), np.mean((p32 - target) ** 2)
    assert l8 < 0.75 * l0 and l32 < 0.75 * l0 and abs(l8 - l32) < 0.02 * l0
    # trajectories agree to a fraction of the distance travelled (20 steps of size ~lr)
    assert np.abs(p8 - p32).max() < 0.15 * 20 * hp["lr"] and np.abs(p8 - p32).mean() < 0.02 * 20 * hp["lr"]
    # the decoded moments are the fp32 moments to within the map's resolution at their magnitude
    m8 = A.dequantize_blockwise(st.state1, st.absmax1, st.qmap1)
    big = np.abs(m) > 0.1 * np.abs(m).max()
    assert np.all(np.abs(m8[big] - m[big]) < 0.05 * np.abs(m).max())


def test_get_optimizer_8bit_on_cpu_keeps_the_references_import_error():
    m = torch.nn.Linear(8, 8)
    with pytest.raises(ImportError, match="bitsandbytes"):
        wopt.get_optimizer(m, {"type": "adamw", "8bit": True, "params": {"lr": 1e-3}})


def test_state_dict_round_trip_keeps_uint8_codes_and_shared_maps():
    """torch's Optimizer.load_state_dict casts state tensors to the parameter dtype — the code bytes came back float32 and the
    next step refused them.  (State built by hand: the update itself is a HIP kernel.)"""
    p = torch.nn.Parameter(torch.randn(8192))
    small = torch.nn.Parameter(torch.randn(16))
    opt = wopt.WftAdamW8bit([p, small], lr=1e-3)
    g = torch.Generator().manual_seed(0)
    q1, q2 = opt._maps(p.device)
    opt.state[p] = {"step": 3, "state1": torch.randint(0, 256, (8192,), dtype=torch.uint8, generator=g),
                    "state2": torch.randint(0, 256, (8192,), dtype=torch.uint8, generator=g),
                    "absmax1": torch.rand(4, generator=g), "absmax2": torch.rand(4, generator=g), "qmap1": q1, "qmap2": q2}
    opt.state[small] = {"step": 3, "exp_avg": torch.randn(16, generator=g), "exp_avg_sq": torch.rand(16, generator=g)}
    import copy

    sd = copy.deepcopy(opt.state_dict())
    opt2 = wopt.WftAdamW8bit([torch.nn.Parameter(p.detach().clone()), torch.nn.Parameter(small.detach().clone())], lr=1e-3)
    opt2.load_state_dict(sd)
    p2, s2 = opt2.param_groups[0]["params"]
    st = opt2.state[p2]
    assert st["state1"].dtype == torch.uint8 and st["state2"].dtype == torch.uint8 and st["state1"].is_contiguous()
    assert torch.equal(st["state1"], opt.state[p]["state1"]) and torch.equal(st["state2"], opt.state[p]["state2"])
    assert torch.equal(st["absmax1"], opt.state[p]["absmax1"]) and st["absmax1"].dtype == torch.float32
    assert st["qmap1"] is opt2._maps(p2.device)[0] and st["qmap2"] is opt2._maps(p2.device)[1]
    assert st["step"] == 3 and isinstance(st["step"], int)
    assert torch.equal(opt2.state[s2]["exp_avg"], opt.state[small]["exp_avg"])


# ------------------------------------------------------------------------------------------------------------------ GPU
def _gpu_case(n, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** float(torch.empty(1).uniform_(-3, 0, generator=g))) for _ in range(4)]
    return p, grads


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2048, 4096 + 1000, 65536 + 2048 * 3 + 5, 3 * 65536])
def test_kernel_matches_the_restatement(n):
    """wft_mt_adamw8 against oracle/adam8bit_oracle.py over four steps (chunk and block tails, two tensors in one launch, the clip
    coefficient folded in): parameters to fp32 rounding; codes identical except where fused-multiply-add contraction moves a value
    across a midpoint (allowed: |code difference| <= 1 on < 0.5 % of the elements); absmax to 1e-5 relative (measured: 1.1e-6)."""
    from whisper_finetune.engine import kernels as K

    dev = torch.device("cuda:0")
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.1)
    cases = [_gpu_case(n, 0), _gpu_case(max(n // 2, 2048), 1)]
    ps = [c[0].clone().to(dev) for c in cases]
    nbs = [(p.numel() + 2047) // 2048 for p in ps]
    s1 = [torch.zeros(p.numel(), dtype=torch.uint8, device=dev) for p in ps]
    s2 = [torch.zeros(p.numel(), dtype=torch.uint8, device=dev) for p in ps]
    a1 = [torch.zeros(nb, device=dev) for nb in nbs]
    a2 = [torch.zeros(nb, device=dev) for nb in nbs]
    q1, q2 = wopt.create_dynamic_map(True).to(dev), wopt.create_dynamic_map(False).to(dev)
    table = K.TensorTable(ps)
    ref_p = [c[0].numpy().copy() for c in cases]
    ref_st = [A.Adam8bitState(p.size) for p in ref_p]
    for step in range(1, 5):
        gs = [c[1][step - 1] for c in cases]
        sumsq = torch.tensor([sum(float((g.double() ** 2).sum()) for g in gs)], device=dev, dtype=torch.float32)
        max_norm = 0.5 * float(sumsq.sqrt())  # clip coefficient ~0.5
        coef = min(1.0, max_norm / (float(sumsq.sqrt()) + 1e-6))
        K.mt_adamw8(table, ps, [g.to(dev) for g in gs], s1, s2, a1, a2, q1, q2, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"],
                    hp["weight_decay"], 1 - hp["beta1"] ** step, 1 - hp["beta2"] ** step, sumsq, max_norm)
        torch.cuda.synchronize()
        for i in range(2):
            ref_p[i] = A.adamw8bit_step(ref_p[i], gs[i].numpy(), ref_st[i], gnorm_scale=np.float32(coef), **hp)
            got_p = ps[i].cpu().numpy()
            assert np.abs(got_p - ref_p[i]).max() <= 2e-6 * max(1.0, np.abs(ref_p[i]).max()), (step, i)
            for got, want in ((s1[i], ref_st[i].state1), (s2[i], ref_st[i].state2)):
                d = np.abs(got.cpu().numpy().astype(int) - want.astype(int))
                assert d.max() <= 1 and (d > 0).mean() < 5e-3, (step, i, d.max(), (d > 0).mean())
            np.testing.assert_allclose(a1[i].cpu().numpy(), ref_st[i].absmax1, rtol=1e-5)
            np.testing.assert_allclose(a2[i].cpu().numpy(), ref_st[i].absmax2, rtol=1e-5)
            # keep the two trajectories on the same state: a flipped code is a legitimate nearest-neighbour tie, not drift to chase
            ref_st[i].state1, ref_st[i].state2 = s1[i].cpu().numpy().copy(), s2[i].cpu().numpy().copy()
            ref_st[i].absmax1, ref_st[i].absmax2 = a1[i].cpu().numpy().copy(), a2[i].cpu().numpy().copy()
            ref_p[i] = got_p.copy()


@pytest.mark.gpu
def test_get_optimizer_honours_8bit_and_trains_like_fp32_adamw():
    """`optimizer.8bit: True` (configs/config_turbo_best.yaml:66) builds WftAdamW8bit with bitsandbytes' state layout; small tensors keep
    fp32 moments; thirty steps on a small regression follow WftAdamW's loss curve."""
    dev = torch.device("cuda:0")
    conf = {"type": "adamw", "8bit": True, "params": {"lr": 3e-3, "weight_decay": 0.1, "betas": [0.9, 0.98], "eps": 1e-6}}

    def run(eight: bool):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(96, 128), torch.nn.Tanh(), torch.nn.Linear(128, 64)).to(dev)
        opt = wopt.get_optimizer(net, {**conf, "8bit": eight})
        x = torch.randn(512, 96, device=dev)
        y = torch.randn(512, 64, device=dev)
        losses = []
        for _ in range(30):
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(net(x), y)
            loss.backward()
            opt.fuse_clip_grad_norm(1.0)
            opt.step()
            losses.append(loss.item())
        return opt, net, losses

    o8, n8, l8 = run(True)
    o32, n32, l32 = run(False)
    assert isinstance(o8, wopt.WftAdamW8bit) and isinstance(o32, wopt.WftAdamW)
    st_w, st_b = o8.state[n8[0].weight], o8.state[n8[0].bias]
    assert set(st_w) == {"step", "state1", "state2", "absmax1", "absmax2", "qmap1", "qmap2"} and st_w["state1"].dtype == torch.uint8
    assert st_w["absmax1"].numel() == (96 * 128 + 2047) // 2048 and set(st_b) == {"step", "exp_avg", "exp_avg_sq"}  # 128 < 4 096
    assert l8[-1] < 0.9 * l8[0] and l32[-1] < 0.9 * l32[0]
    assert all(abs(a - b) < 0.02 * l32[0] for a, b in zip(l8, l32))
    sd = o8.state_dict()  # the state round-trips through the optimizer's own (torch) state_dict
    o8.load_state_dict(sd)
