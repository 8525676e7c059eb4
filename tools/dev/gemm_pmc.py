"""Developer script (GPU box): a handful of launches of the two 256x256 GEMM kernels at one encoder shape on random data, meant to
run under `rocprofv3 --pmc ...` (one counter group per pass; summarise with tools/dev/pmc_sum.py):
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmc_lds -- python3 tools/dev/gemm_pmc.py
Without a profiler it prints the two launch times."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, Kd = 102000, 5120, 1280
x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
w = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dw = torch.empty(N, Kd, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, fn in (("nt256", lambda: K.gemm_nt(x, w, out=y)), ("tn256", lambda: K.gemm_tn(dy, x, out=dw))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print(f"{name}: {t*1e3:.3f} ms  {2.0*M*N*Kd/t/1e12:.0f} TFLOP/s", flush=True)
