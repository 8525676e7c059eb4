"""Diagnostic: per-slab s_memtime stamps of gemm_nt256_kernel (library built with -DWFT_STAMPS into libwft_stamps.so)."""
import ctypes as C, sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
from whisper_finetune.engine import lib as L
L.LIB_PATH = ROOT / "whisper-finetune_amd" / "libwft_stamps.so"
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
M, N, Kd = 48000, 5120, 1280
a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ws = torch.zeros(2 * 64 * 6, dtype=torch.int64, device=dev)
args = L.GemmArgs()
args.A, args.lda, args.B, args.ldb, args.C, args.ldc = a.data_ptr(), Kd, b.data_ptr(), Kd, out.data_ptr(), N
args.M, args.N, args.K, args.batch, args.alpha = M, N, Kd, 1, 1.0
args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(5):
    L.check(L.load().wft_gemm_nt_bf16(C.byref(args), L.stream_ptr()), "gemm")
torch.cuda.synchronize()
st = ws.cpu().numpy().reshape(2, 64, 6)[:, :40].astype(np.int64)
for g, name in ((0, "group A (wave 0)"), (1, "group B (wave 4)")):
    s = st[g]
    d = np.diff(s, axis=1)            # within-slab segments 0-1 issue reads+glds, 1-2 lgkm wait, 2-3 vmcnt wait, 3-4 barrier1, 4-5 C-unit
    nxt = s[1:, 0] - s[:-1, 5]        # barrier2 + loop overhead
    per = s[1:, 0] - s[:-1, 0]
    mid = slice(5, 35)
    print(name, "period", per[mid].mean().round(1), "| issue", d[mid, 0].mean().round(1), "lgkm", d[mid, 1].mean().round(1), "vmcnt", d[mid, 2].mean().round(1),
          "bar1", d[mid, 3].mean().round(1), "C-unit", d[mid, 4].mean().round(1), "bar2", nxt[mid].mean().round(1))
