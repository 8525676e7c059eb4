"""Developer script (GPU box): the hand-written bf16 GEMMs against hipBLASLt (torch.matmul on ROCm dispatches to it) on the
large-v3 encoder shapes, both in ONE process on random data, as a markdown table.
    python tools/gemm_vs_hipblaslt.py > gpurun_out/r02_gemm_vs_hipblaslt.md
hipBLASLt is the yard-stick only: the product path never calls it."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
bf = lambda x: x.to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
print("| kernel | shape (M x N x K) | libwft ms | libwft TF/s | hipBLASLt ms | hipBLASLt TF/s | libwft / hipBLASLt |")
print("|---|---|---|---|---|---|---|")
for M in (102000, 48000):  # 68 and 32 clips x 1500 frames
    for N, Kd in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
        a = bf(torch.randn(M, Kd, device=dev)); b = bf(torch.randn(N, Kd, device=dev))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: K.gemm_nt(a, b, out=out))
        th = timeit(lambda: torch.matmul(a, b.t(), out=out))
        fl = 2.0 * M * N * Kd
        print(f"| NT (forward / backward-data) | {M} x {N} x {Kd} | {t*1e3:.3f} | {fl/t/1e12:.0f} | {th*1e3:.3f} | {fl/th/1e12:.0f} | {th/t:.2f} |", flush=True)
    for P, Q in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
        a = bf(torch.randn(M, P, device=dev)); b = bf(torch.randn(M, Q, device=dev))
        out = torch.empty(P, Q, device=dev)
        outh = torch.empty(P, Q, dtype=torch.float32, device=dev)
        t = timeit(lambda: K.gemm_tn(a, b, out=out))
        a32t = a.t()
        th = timeit(lambda: torch.matmul(a32t, b, out=torch.empty(P, Q, dtype=torch.bfloat16, device=dev)))
        fl = 2.0 * M * P * Q
        print(f"| TN (weight gradient, fp32 out, deterministic split-K) | R={M}: {P} x {Q} | {t*1e3:.3f} | {fl/t/1e12:.0f} | {th*1e3:.3f} (bf16 out) | {fl/th/1e12:.0f} | {th/t:.2f} |", flush=True)
