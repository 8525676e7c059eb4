import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, Kd = 48000, 5120, 1280
a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    K.gemm_nt(a, b, out=out)
if len(sys.argv) > 1:
    a2 = torch.randn(M, N, device=dev).to(torch.bfloat16); b2 = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    o2 = torch.empty(N, Kd, device=dev)
    for _ in range(5):
        K.gemm_tn(a2, b2, out=o2)
torch.cuda.synchronize()
