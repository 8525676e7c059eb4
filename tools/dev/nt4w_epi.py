"""Developer script (GPU box): what the epilogue variants of gemm_nt4w_kernel cost on the fc1 / fc2-dgrad shapes (timing builds via
WFT_GEMM_DIAG: 22 C stores dropped, 23 C and aux stores dropped, 24 aux stores dropped).  One process per setting:
    for d in 0 22 23 24; do WFT_GEMM_DIAG=$d python tools/dev/nt4w_epi.py; done"""
import os, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__))); import _timing_lib  # noqa: E702 (WFT_LIB -> libwft_timing.so)
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 102000
out = []
for name, N, Kd in (("fc1", 5120, 1280), ("fc2 dgrad", 5120, 1280)):
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.03).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.randn(M, N, device=dev).to(torch.bfloat16); bias = torch.randn(N, device=dev)
    cs = torch.empty(N, device=dev)
    fl = 2.0 * M * N * Kd
    if name == "fc1":
        t0 = min(timeit(lambda: K.gemm_nt(a, b, out=c, bias=bias)) for _ in range(3))
        t1 = min(timeit(lambda: K.gemm_nt(a, b, out=c, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)) for _ in range(3))
        out.append(f"fc1 plain+bias {t0*1e6:.0f} us ({fl/t0/1e12:.0f} TF/s) GELU_GRAD {t1*1e6:.0f} us ({fl/t1/1e12:.0f})")
    else:
        t1 = min(timeit(lambda: K.gemm_nt(a, b, out=c, epilogue=L.EPI_MUL_AUX, aux=aux)) for _ in range(3))
        t2 = min(timeit(lambda: K.gemm_nt(a, b, out=c, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=cs)) for _ in range(3))
        out.append(f"MUL_AUX {t1*1e6:.0f} us ({fl/t1/1e12:.0f}) + colsum {t2*1e6:.0f} us ({fl/t2/1e12:.0f})")
for name, N, Kd in (("out-proj + residual", 1280, 1280), ("fc2 + residual", 1280, 5120)):  # the <NONE, residual> variant
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.03).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev); res = torch.randn(M, N, device=dev).to(torch.bfloat16); bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * Kd
    t0 = min(timeit(lambda: K.gemm_nt(a, b, out=c, bias=bias)) for _ in range(3))
    t1 = min(timeit(lambda: K.gemm_nt(a, b, out=c, bias=bias, residual=res)) for _ in range(3))
    out.append(f"{name}: bias only {t0*1e6:.0f} us ({fl/t0/1e12:.0f}) + residual {t1*1e6:.0f} us ({fl/t1/1e12:.0f})")
print(f"{os.path.basename(os.environ.get('WFT_LIB', 'libwft.so'))} DIAG {os.environ.get('WFT_GEMM_DIAG', '0')}: " + " | ".join(out), flush=True)
