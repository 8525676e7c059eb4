"""dev tool: persistent NT256 GEMM correctness / determinism stress for the counted-wait epilogue (asm ring loads)."""
import sys, torch
sys.path.insert(0, "whisper-finetune_amd")
from whisper_finetune.engine import kernels as K, lib as L

dev = "cuda"
torch.manual_seed(0)
bad = 0
for M, N, Kd in ((20000, 5120, 1280), (102000, 1280, 1280), (30000, 1280, 5120), (12345, 2560, 192), (9000, 3840, 256), (70001, 1280, 128)):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16(); aux = torch.randn(M, N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    ref = (a.float() @ b.float().t())
    cases = {
        "bias": (lambda: K.gemm_nt(a, b, bias=bias), ref + bias),
        "bias+res": (lambda: K.gemm_nt(a, b, bias=bias, residual=res), ref + bias + res.float()),
        "res": (lambda: K.gemm_nt(a, b, residual=res, beta=0.5), ref + 0.5 * res.float()),
        "mulaux": (lambda: K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=aux), ref * aux.float()),
        "mulaux+cs": (lambda: K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=torch.empty(N, device=dev)), ref * aux.float()),
        "gelu_grad": (lambda: K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=torch.empty_like(res)), torch.nn.functional.gelu(ref + bias)),
    }
    for name, (fn, want) in cases.items():
        first = fn()
        err = (first.float() - want).abs().max().item() / want.abs().max().item()
        nd = 0
        for _ in range(10):
            nd += int(not torch.equal(fn(), first))
        flag = "" if (err < 1e-2 and nd == 0) else "  <-- BAD"
        bad += bool(flag)
        print(f"M={M} N={N} K={Kd} {name:10s} rel err {err:.2e}  non-identical repeats {nd}{flag}", flush=True)
print("BAD" if bad else "ALL OK")
