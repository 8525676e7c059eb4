"""Developer script (GPU box): dQ kernels A/B (wft_attn_set_dq_variant 0 = one wave per SIMD, 1 = 8-wave) — results against each
other and against fp32 math, then timings.   python tools/dev/dq4w_ab.py"""
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

def ref_dq(q, k, v, do, H, scale):
    B, Tq, _ = q.shape; Tk = k.shape[1]
    qf, kf, vf, dof = (x.float().view(B, -1, H, 64).transpose(1, 2) for x in (q, k, v, do))
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1)
    dp = dof @ vf.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    return (ds @ kf * scale).transpose(1, 2).reshape(B, Tq, H * 64)

rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
for B, H, Tq, Tk in ((2, 8, 1500, 1500), (1, 8, 600, 70), (2, 8, 513, 333), (1, 5, 777, 257), (2, 20, 1500, 1500), (1, 8, 512, 64)):
    for rep in range(2):
        qkv = (torch.randn(B, Tq, 3 * H * 64, device=dev, generator=g)).to(torch.bfloat16)
        q = qkv[..., :H * 64]
        kv = torch.randn(B, Tk, 2 * H * 64, device=dev, generator=g).to(torch.bfloat16)
        k, v = kv[..., :H * 64], kv[..., H * 64:]
        do = torch.randn(B, Tq, H * 64, device=dev, generator=g).to(torch.bfloat16)
        o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
        outs = []
        for var in (1, 0):
            K.set_variant("dq", var)
            cs = (torch.empty(H * 64, device=dev), torch.empty(H * 64, device=dev))
            dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
            torch.cuda.synchronize()
            outs.append((dq.clone(), dk.clone(), dv.clone(), cs[0].clone()))
        rq = ref_dq(q, k, v, do, H, 0.125)
        print(f"B{B} H{H} {Tq}x{Tk} rep{rep}: dq 4w-vs-8w {rel(outs[1][0], outs[0][0]):.2e} dk {rel(outs[1][1], outs[0][1]):.2e} dv {rel(outs[1][2], outs[0][2]):.2e} cs_q {rel(outs[1][3], outs[0][3]):.2e}"
              f" | vs fp32: dq 8w {rel(outs[0][0], rq):.2e} 4w {rel(outs[1][0], rq):.2e} nan {int(torch.isnan(outs[1][0].float()).sum())}", flush=True)
for B, H, Tq, Tk in ((32, 20, 1500, 1500),):
    q = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * H * 64, device=dev).to(torch.bfloat16)
    k, v = kv[..., :H * 64], kv[..., H * 64:]
    do = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    for rnd in range(2):
        for var in (1, 0):
            K.set_variant("dq", var)
            ms = t(lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125))
            print(f"B{B} {Tq}x{Tk} dq variant {var}: bwd {ms:.3f} ms", flush=True)
