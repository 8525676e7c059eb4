"""Developer script: GEMM correctness + perf at the large-v3 shapes (GPU box)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
from whisper_finetune.engine import lib as L
dev = torch.device("cuda:0")
def bf(x): return x.to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
for (M, N, Kd) in ((48000, 1280, 1280), (48000, 3840, 1280), (48000, 5120, 1280), (48000, 1280, 5120), (4096, 51968, 1280), (4096, 1280, 1280), (47999, 1280, 1280)):
    a = bf(torch.randn(M, Kd, device=dev)); b = bf(torch.randn(N, Kd, device=dev))
    bias = torch.randn(N, device=dev); res = bf(torch.randn(M, N, device=dev))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ref = torch.matmul(a, b.t()).float()
    got = K.gemm_nt(a, b, out=out).float()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    aux = torch.empty_like(out)
    g2 = K.gemm_nt(a, b, bias=bias, residual=res, epilogue=L.EPI_GELU, aux=aux).float()
    r2 = torch.nn.functional.gelu(ref + bias) + res.float()
    err2 = (g2 - r2).abs().max().item() / r2.abs().max().item()
    t = timeit(lambda: K.gemm_nt(a, b, out=out))
    tg = timeit(lambda: K.gemm_nt(a, b, out=out, bias=bias, epilogue=L.EPI_GELU, aux=aux))
    th = timeit(lambda: torch.matmul(a, b.t(), out=out))
    print(f"nt {M}x{N}x{Kd}: err {err:.2e} gelu-err {err2:.2e} | {t*1e3:.3f} ms {2*M*N*Kd/t/1e12:.0f} TF/s | gelu-epi {2*M*N*Kd/tg/1e12:.0f} | hipblaslt {2*M*N*Kd/th/1e12:.0f} TF/s", flush=True)
for (R, P, Q) in ((48000, 1280, 1280), (48000, 3840, 1280), (48000, 5120, 1280), (48000, 1280, 5120), (4096, 1280, 1280)):
    a = bf(torch.randn(R, P, device=dev)); b = bf(torch.randn(R, Q, device=dev))
    out = torch.empty(P, Q, device=dev)
    ref = torch.matmul(a.t(), b).float()
    got = K.gemm_tn(a, b, out=out)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(lambda: K.gemm_tn(a, b, out=out))
    outb = torch.empty(P, Q, dtype=torch.bfloat16, device=dev)
    th = timeit(lambda: torch.matmul(a.t(), b, out=outb))
    print(f"tn {R}:{P}x{Q}: err {err:.2e} | {t*1e3:.3f} ms {2*R*P*Q/t/1e12:.0f} TF/s | hipblaslt {2*R*P*Q/th/1e12:.0f} TF/s", flush=True)
