"""A/B helper: time the main GEMM shapes several times in one process (variance check)."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = ((48000, 1280, 1280), (48000, 3840, 1280), (48000, 5120, 1280), (48000, 1280, 5120))
data = []
for (M, N, Kd) in shapes:
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    data.append((a, b, torch.empty(M, N, dtype=torch.bfloat16, device=dev)))
for rnd in range(3):
    res = []
    for (M, N, Kd), (a, b, out) in zip(shapes, data):
        for _ in range(3): K.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): K.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        res.append(f"{2*M*N*Kd/t/1e12:.0f}")
    print("round", rnd, "NT TF/s:", " ".join(res), flush=True)
