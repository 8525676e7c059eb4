"""dev tool: cost of the NT256 epilogue variants at the encoder MLP shapes (B x 1500 rows)."""
import sys, torch
sys.path.insert(0, "whisper-finetune_amd")
from whisper_finetune.engine import kernels as K, lib as L

def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 68
M = B * 1500
dev = "cuda"
for N, Kd in ((5120, 1280), (1280, 5120), (1280, 1280), (3840, 1280)):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = (torch.randn(N, Kd, device=dev) * 0.03).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux = torch.randn(M, N, device=dev).bfloat16(); res = torch.randn(M, N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); cs = torch.empty(N, device=dev)
    fl = 2.0 * M * N * Kd
    cases = {
        "none": lambda: K.gemm_nt(a, b, out=out),
        "bias": lambda: K.gemm_nt(a, b, out=out, bias=bias),
        "bias+res": lambda: K.gemm_nt(a, b, out=out, bias=bias, residual=res),
        "gelu+aux": lambda: K.gemm_nt(a, b, out=out, bias=bias, epilogue=L.EPI_GELU, aux=aux),
        "gelu_grad": lambda: K.gemm_nt(a, b, out=out, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux),
        "dgelu": lambda: K.gemm_nt(a, b, out=out, epilogue=L.EPI_DGELU, aux=aux),
        "dgelu+cs": lambda: K.gemm_nt(a, b, out=out, epilogue=L.EPI_DGELU, aux=aux, colsum=cs),
        "mulaux": lambda: K.gemm_nt(a, b, out=out, epilogue=L.EPI_MUL_AUX, aux=aux),
        "mulaux+cs": lambda: K.gemm_nt(a, b, out=out, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=cs),
    }
    for name, fn in cases.items():
        us = t(fn)
        print(f"M={M} N={N} K={Kd} {name:10s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s", flush=True)
