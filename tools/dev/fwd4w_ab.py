"""Developer script (GPU box): forward attention kernels A/B (wft_attn_set_fwd_variant 0 = one wave per SIMD, 1 = 8-wave) — outputs and
lse against fp32 math and against each other, then timings.   python tools/dev/fwd4w_ab.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
lib = L.load()
g = torch.Generator(device=dev).manual_seed(0)

def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def ref(q, k, v, H, scale):
    B, Tq, _ = q.shape
    qf, kf, vf = (x.float().view(B, -1, H, 64).transpose(1, 2) for x in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * scale
    return (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Tq, H * 64), torch.logsumexp(s, -1)

rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
for B, H, Tq, Tk, amp in ((2, 8, 1500, 1500, 1.0), (1, 8, 600, 70, 1.0), (2, 8, 513, 333, 3.0), (1, 5, 777, 257, 1.0), (2, 20, 1500, 1500, 4.0), (1, 8, 512, 64, 1.0), (1, 8, 640, 1, 1.0)):
    for rep in range(2):
        qkv = (torch.randn(B, Tq, 3 * H * 64, device=dev, generator=g) * amp).to(torch.bfloat16)
        q = qkv[..., :H * 64]
        kv = (torch.randn(B, Tk, 2 * H * 64, device=dev, generator=g) * amp).to(torch.bfloat16)
        k, v = kv[..., :H * 64], kv[..., H * 64:]
        outs = []
        for var in (1, 0):
            lib.wft_attn_set_fwd_variant(var)
            o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
            torch.cuda.synchronize()
            outs.append((o.clone(), lse.clone()))
        ro, rl = ref(q, k, v, H, 0.125)
        print(f"B{B} H{H} {Tq}x{Tk} amp{amp} rep{rep}: o 4w-vs-8w {rel(outs[1][0], outs[0][0]):.2e} lse max diff {(outs[1][1] - outs[0][1]).abs().max().item():.2e}"
              f" | vs fp32: o 8w {rel(outs[0][0], ro):.2e} 4w {rel(outs[1][0], ro):.2e}  lse 8w {(outs[0][1] - rl).abs().max().item():.2e} 4w {(outs[1][1] - rl).abs().max().item():.2e}"
              f" nan {int(torch.isnan(outs[1][0].float()).sum())}", flush=True)
for B, H, Tq, Tk in ((32, 20, 1500, 1500),):
    q = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * H * 64, device=dev).to(torch.bfloat16)
    k, v = kv[..., :H * 64], kv[..., H * 64:]
    for rnd in range(2):
        for var in (1, 0):
            lib.wft_attn_set_fwd_variant(var)
            ms = t(lambda: K.attn_fwd(q, k, v, H, False, 0.125))
            print(f"B{B} {Tq}x{Tk} fwd variant {var}: {ms:.3f} ms", flush=True)
