"""Developer script (GPU box): time one NT shape for the current environment (A/B of env switches across processes is
unreliable; this is for coarse effects only).  python tools/dev/nt4w_one.py M N K [reps]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
M, N, Kd = [int(x) for x in sys.argv[1:4]]
torch.manual_seed(0)
a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.03).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bias = torch.randn(N, device=dev)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
ts = sorted(timeit(lambda: K.gemm_nt(a, b, out=out, bias=bias)) for _ in range(5))
print(f"{M}x{N}x{Kd}: {ts[2]*1e6:.1f} us  {2.0*M*N*Kd/ts[2]/1e12:.0f} TF/s (min {ts[0]*1e6:.1f})", flush=True)
