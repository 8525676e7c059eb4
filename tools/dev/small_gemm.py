import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M in (8704, 4096):
    for N, Kd in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120), (2560, 1280)):
        a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
        us = bench(lambda: K.gemm_nt(a, b))
        print(f"NT M{M} N{N} K{Kd}: {us:7.1f} us {2*M*N*Kd/us/1e6:7.1f} TF/s")
    for P, Q in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
        a = torch.randn(M, P, device=dev).to(torch.bfloat16); b = torch.randn(M, Q, device=dev).to(torch.bfloat16)
        us = bench(lambda: K.gemm_tn(a, b))
        print(f"TN R{M} P{P} Q{Q}: {us:7.1f} us {2*M*P*Q/us/1e6:7.1f} TF/s")
