import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, Kd) in ((48000, 5120, 1280), (48000, 1280, 5120)):
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3): K.gemm_nt(a, b, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): K.gemm_nt(a, b, out=out)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(f"{M}x{N}x{Kd}: {t*1e3:.3f} ms {2*M*N*Kd/t/1e12:.0f} TF/s")
