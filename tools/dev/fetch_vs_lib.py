"""Developer script (GPU box, under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`): L2-miss fetch traffic of gemm_nt256 and of the
vendor library's kernel on the four encoder shapes (one launch each after warm-up; parse the counter CSV per kernel)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
M = 102000
for N, Kd in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev); bt = b.t()
    for _ in range(3):
        K.gemm_nt(a, b, out=o)
        torch.mm(a, bt, out=o)
    torch.cuda.synchronize()
