"""Developer script (GPU box): dK/dV kernels A/B (wft_attn_set_dkdv_variant 0 = one wave per SIMD, 1 = 8-wave) — results against
each other and against fp32 math, then timings.   python tools/dev/dkdv4w_ab.py"""
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

def ref_bwd(q, k, v, do, H, scale):
    B, Tq, _ = q.shape; Tk = k.shape[1]
    qf = q.float().view(B, Tq, H, 64).transpose(1, 2); kf = k.float().view(B, Tk, H, 64).transpose(1, 2)
    vf = v.float().view(B, Tk, H, 64).transpose(1, 2); dof = do.float().view(B, Tq, H, 64).transpose(1, 2)
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1)
    dv = p.transpose(-1, -2) @ dof
    dp = dof @ vf.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    dk = ds.transpose(-1, -2) @ qf * scale
    return dk.transpose(1, 2).reshape(B, Tk, H * 64), dv.transpose(1, 2).reshape(B, Tk, H * 64)

rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
for B, H, Tq, Tk in ((2, 8, 1500, 1500), (3, 6, 200, 1500), (1, 8, 128, 70), (2, 8, 449, 333), (2, 20, 1500, 1500)):
    for rep in range(2):
        qkv = (torch.randn(B, Tq, 3 * H * 64, device=dev, generator=g)).to(torch.bfloat16)
        q = qkv[..., :H * 64]
        kv = torch.randn(B, Tk, 2 * H * 64, device=dev, generator=g).to(torch.bfloat16)
        k, v = kv[..., :H * 64], kv[..., H * 64:]
        do = torch.randn(B, Tq, H * 64, device=dev, generator=g).to(torch.bfloat16)
        o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
        outs = []
        for var in (1, 0):
            K.set_variant("dkdv", var)
            cs = (torch.empty(H * 64, device=dev), torch.empty(H * 64, device=dev))
            dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
            torch.cuda.synchronize()
            outs.append((dk.clone(), dv.clone(), cs[1].clone()))
        rk, rv = ref_bwd(q, k, v, do, H, 0.125)
        print(f"B{B} H{H} {Tq}x{Tk} rep{rep}: dk 4w-vs-8w {rel(outs[1][0], outs[0][0]):.2e} dv {rel(outs[1][1], outs[0][1]):.2e} cs_v {rel(outs[1][2], outs[0][2]):.2e}"
              f" | vs fp32: dk 8w {rel(outs[0][0], rk):.2e} 4w {rel(outs[1][0], rk):.2e}  dv 8w {rel(outs[0][1], rv):.2e} 4w {rel(outs[1][1], rv):.2e}"
              f" nan {int(torch.isnan(outs[1][0].float()).sum())} {int(torch.isnan(outs[1][1].float()).sum())}", flush=True)
for B, H, Tq, Tk in ((32, 20, 1500, 1500), (64, 20, 128, 1500)):
    q = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * H * 64, device=dev).to(torch.bfloat16)
    k, v = kv[..., :H * 64], kv[..., H * 64:]
    do = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    for rnd in range(2):
        for var in (1, 0):
            K.set_variant("dkdv", var)
            ms = t(lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125))
            print(f"B{B} {Tq}x{Tk} variant {var}: bwd {ms:.3f} ms", flush=True)
