"""Developer script (GPU box): where one headline step's kernel time goes BY SHAPE — every libwft wrapper call of one steady-state
train_step bracketed by HIP events on the launch stream (serialised: the numbers are per-call device times, not overlapped time).
    python tools/dev/shape_times.py [batch] [model]"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import bench  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402


class A:
    model = sys.argv[2] if len(sys.argv) > 2 else "large-v3"


B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
case = bench.Case(A, dev, 0, 0, 1, False)
case.measure(B, 128, 1, 2, roofline=False)
rec = []


def wrap(name, keyf, flopf):
    real = getattr(K, name)

    def f(*a, **kw):
        if kw.get("_args_only"):
            return real(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real(*a, **kw)
        e1.record()
        rec.append((name, keyf(*a, **kw), flopf(*a, **kw), e0, e1))
        return out

    setattr(K, name, f)


def nt_key(a, b, **kw):
    M = kw.get("M") or a.shape[0]
    return (M * kw.get("batch", 1), kw.get("N") or b.shape[0], kw.get("K") or a.shape[-1], kw.get("epilogue", 0), kw.get("residual") is not None)


def tn_key(a, b, **kw):
    return ((kw.get("R") or a.shape[0]) * kw.get("batch", 1), kw.get("P") or a.shape[1], kw.get("Q") or b.shape[1], kw.get("p_valid", 0))


wrap("gemm_nt", nt_key, lambda a, b, **kw: 2.0 * nt_key(a, b, **kw)[0] * nt_key(a, b, **kw)[1] * nt_key(a, b, **kw)[2])
wrap("gemm_tn", tn_key, lambda a, b, **kw: 2.0 * tn_key(a, b, **kw)[0] * tn_key(a, b, **kw)[1] * tn_key(a, b, **kw)[2])
wrap("attn_fwd", lambda q, k, v, H, causal, scale, **kw: (q.shape[0], H, q.shape[1], k.shape[1], bool(causal)),
     lambda q, k, v, H, causal, scale, **kw: 4.0 * q.shape[0] * H * q.shape[1] * k.shape[1] * 64 * (0.5 if causal else 1))
wrap("attn_bwd", lambda q, k, v, o, lse, do, H, causal, scale, **kw: (q.shape[0], H, q.shape[1], k.shape[1], bool(causal)),
     lambda q, k, v, o, lse, do, H, causal, scale, **kw: 10.0 * q.shape[0] * H * q.shape[1] * k.shape[1] * 64 * (0.5 if causal else 1))
for nm in ("layernorm_fwd", "layernorm_bwd"):
    wrap(nm, lambda x, *a, **kw: (x.shape[0], x.shape[1]), lambda x, *a, **kw: 0.0)

t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
case.measure(B, 128, 1, 0, roofline=False)
t1.record()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, key, fl, e0, e1 in rec:
    r = agg[(name, key)]
    r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += fl
tot = sum(r[1] for r in agg.values())
print(f"batch {B}: wrapped calls {len(rec)}, their device time {tot:.1f} ms (instrumented step {t0.elapsed_time(t1):.1f} ms)")
for (name, key), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.2f} ms {n:5d} x {ms / n * 1e3:8.1f} us  {fl / ms / 1e9 if ms > 0 and fl > 0 else 0:7.0f} TF/s  {name} {key}")
