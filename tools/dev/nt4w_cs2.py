import sys, ctypes as C
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
for rep in range(6):
  for (M, N, Kd) in ((1024, 256, 768), (2048 + 112, 512, 896), (4000, 1280, 1280), (3072, 256, 5120), (1500 * 3, 3840, 1280)):
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(torch.bfloat16); auxin = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ref0 = a.float() @ b.float().t()
    if rep > 0:
        K.gemm_nt(a, b, bias=bias); K.gemm_nt(a, b, bias=bias, residual=res)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev); K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)
    cs = torch.full((N,), float("nan"), device=dev)
    out = K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=auxin, colsum=cs)
    torch.cuda.synchronize()
    refcs = (ref0 * auxin.float()).sum(0)
    d = (cs - refcs).abs()
    bad = (~(d < 1e-2 * refcs.abs().max())).nonzero().flatten()
    if bad.numel():
        ws = K._TN_WS[("cuda", 0, "nt_colsum")][: 2 * ((M + 255) // 256) * N * 4].view(torch.float32).view(-1, N)
        nanrows = torch.isnan(ws).any(1).nonzero().flatten().tolist()
        print("   partial rows with NaN:", nanrows, "of", ws.shape[0], "cols with NaN in first such row:", torch.isnan(ws[nanrows[0]]).nonzero().flatten()[:40].tolist() if nanrows else None)
        ref = ref0 * auxin.float()
        dd = (out.float() - ref).abs(); badr = (dd > 0.05 * ref.abs().max()).any(1).nonzero().flatten()
        print("   out: bad rows", badr.numel(), badr[:20].tolist(), "nan in out:", torch.isnan(out.float()).sum().item())
    print(rep, M, N, Kd, "bad cols", bad.numel(), bad[:16].tolist(), "vals", cs[bad[:4]].tolist(), refcs[bad[:4]].tolist(), flush=True)
