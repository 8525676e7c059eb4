import sys, os, ctypes as C
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
M, N, Kd = [int(x) for x in sys.argv[1:4]]
torch.manual_seed(0)
a = (torch.randn(M, Kd, device=dev)).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.05).to(torch.bfloat16)
out = K.gemm_nt(a, b)
torch.cuda.synchronize()
ref = a.float() @ b.float().t()
err = (out.float() - ref).norm() / ref.norm()
print(M, N, Kd, "rel", err.item(), flush=True)
if err > 1e-2:
    d = (out.float() - ref).abs()
    bad = (d > 0.05 * ref.abs().max()).nonzero()
    print("bad count", bad.shape[0], "first", bad[:10].tolist())
    rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
    print("bad rows", rows[:40].tolist(), "...", rows.numel()); print("bad cols", cols[:40].tolist(), "...", cols.numel())
