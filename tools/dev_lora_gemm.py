import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
M = 48000
for Kd in (1280, 5120):
    x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    a = torch.randn(128, Kd, device=dev).to(torch.bfloat16)
    du = torch.randn(M, 128, device=dev).to(torch.bfloat16)
    bt = torch.randn(128, Kd, device=dev).to(torch.bfloat16)   # BbT [Rpad, N]
    us = bench(lambda: K.gemm_nt(x, a)); print(f"u  = x[{M}x{Kd}] @ Am^T[128]      : {us:7.1f} us  ({M*Kd*2/us/1e6:.2f} TB/s of x)")
    dy = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    us = bench(lambda: K.gemm_nt(dy, bt)); print(f"du = dy[{M}x{Kd}] @ Bb[{Kd}x128]   : {us:7.1f} us")
    us = bench(lambda: K.gemm_tn(du, x)); print(f"dA = du^T[128x{M}] @ x[{M}x{Kd}]   : {us:7.1f} us  ({M*Kd*2/us/1e6:.2f} TB/s of x)")
    us = bench(lambda: K.gemm_tn(dy, du)); print(f"dB = dy^T[{Kd}x{M}] @ u[{M}x128]   : {us:7.1f} us")
