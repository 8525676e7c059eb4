"""Developer script (GPU box): NT256 with a cache-resident working set (batch items re-read the same A rows and rewrite the same C)
against the streaming case of the same tile count: separates memory stalls from issue / clock limits."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, Kd in ((1280, 1280), (5120, 1280), (1280, 5120)):
    Mb, nb = 2048, 50
    a = torch.randn(Mb, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    o = torch.empty(Mb, N, dtype=torch.bfloat16, device=dev)
    t_res = bench(lambda: K.gemm_nt(a, b, out=o, batch=nb, strideA=0, strideB=0, strideC=0))
    A = torch.randn(Mb * nb, Kd, device=dev).bfloat16(); O = torch.empty(Mb * nb, N, dtype=torch.bfloat16, device=dev)
    t_str = bench(lambda: K.gemm_nt(A, b, out=O))
    fl = 2.0 * Mb * nb * N * Kd
    print(f"N={N} K={Kd}: resident {t_res:.1f} us ({fl / t_res / 1e6:.0f} TF/s)   streaming {t_str:.1f} us ({fl / t_str / 1e6:.0f} TF/s)", flush=True)
