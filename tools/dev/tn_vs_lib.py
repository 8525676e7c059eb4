"""Developer script (GPU box): the weight-gradient GEMM (dW[P, Q] = X^T[P, R] dY[R, Q], R = 68 x 1500 rows) through
gemm_tn256 against the vendor library's pick for the same product (torch.mm on the transposed view) — a yard-stick, not a code path."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")

def bench(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

R = 68 * 1500
for P, Q in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
    a = torch.randn(R, P, device=dev).bfloat16(); b = torch.randn(R, Q, device=dev).bfloat16()
    o = torch.zeros(P, Q, dtype=torch.float32, device=dev)
    fl = 2.0 * R * P * Q
    mine = bench(lambda: K.gemm_tn(a, b, out=o))
    lib16 = bench(lambda: torch.mm(a.t(), b))
    try:
        lib32 = bench(lambda: torch.mm(a.t(), b, out_dtype=torch.float32))
    except Exception as e:
        lib32 = float("nan"); print("out_dtype:", str(e)[:100])
    print(f"dW {P}x{Q}: gemm_tn256 {mine:8.1f} us {fl / mine / 1e6:7.1f} TF/s | library bf16-out {lib16:8.1f} us {fl / lib16 / 1e6:7.1f} TF/s | "
          f"library f32-out {lib32:8.1f} us {fl / lib32 / 1e6:7.1f} TF/s", flush=True)
