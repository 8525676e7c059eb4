"""Developer script (GPU box): gemm_nt4w_kernel on the encoder shapes of the headline step, for the library WFT_LIB points at."""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 144000
out = []
for N, Kd in ((1280, 1280), (3840, 1280), (1280, 5120), (5120, 1280)):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    f = lambda: K.gemm_nt(a, b)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(6): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 6 * 1e3)
    out.append(f"{N}x{Kd}: {min(ts):7.1f} us {2.0 * M * N * Kd / min(ts) / 1e6:5.0f} TF/s")
tn = []
for P, Q in ((1280, 5120), (5120, 1280), (3840, 1280), (1280, 1280)):
    a = torch.randn(M, P, device=dev).bfloat16(); b = torch.randn(M, Q, device=dev).bfloat16()
    f = lambda: K.gemm_tn(a, b)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(6): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 6 * 1e3)
    tn.append(f"{P}x{Q}: {min(ts):7.1f} us {2.0 * M * P * Q / min(ts) / 1e6:5.0f} TF/s")
print(f"{os.path.basename(os.environ.get('WFT_LIB', 'libwft.so')):18s} " + " | ".join(out) + "\n" + " " * 15 + "tn " + " | ".join(tn), flush=True)
