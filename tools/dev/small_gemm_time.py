"""Developer script (GPU box): the 128-tile GEMM shapes of a whisper-base step at B = 8 (tools/dev/shape_times.py 8 base), timed alone:
    python tools/dev/small_gemm_time.py            # product dispatch
    WFT_GEMM_DIAG=11 python tools/dev/small_gemm_time.py   # the two-buffer kernels everywhere (A/B)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
NT = [(1024, 512, 512), (1024, 512, 2048), (1024, 2048, 512), (1024, 1536, 512), (1024, 512, 1536), (1024, 512, 51968), (12000, 512, 512),
      (12000, 512, 2048), (12000, 1536, 512), (1024, 51968, 512), (1024, 1280, 51968)]
TN = [(1024, 512, 512), (1024, 512, 2048), (1024, 1536, 512), (1024, 2048, 512), (12000, 512, 512), (12000, 512, 2048), (12000, 1536, 512),
      (12000, 1024, 512), (1024, 51968, 512)]


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "dec32":  # decoder-sized products of large-v3 at B = 32
    NT = [(4096, 1280, 1280), (4096, 1280, 5120), (4096, 1280, 3840), (4096, 1280, 2560), (4096, 2560, 1280)]
    TN = [(4096, 1280, 1280), (4096, 1280, 5120), (4096, 3840, 1280)]
for M, N, Kd in NT:
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    us = t(lambda: K.gemm_nt(a, b))
    print(f"nt {M:6d} {N:6d} {Kd:6d}  {us:8.1f} us  {2.0 * M * N * Kd / us / 1e6:7.0f} TF/s")
for R, P, Q in TN:
    a = torch.randn(R, P, device=dev).bfloat16(); b = torch.randn(R, Q, device=dev).bfloat16()
    us = t(lambda: K.gemm_tn(a, b))
    print(f"tn {R:6d} {P:6d} {Q:6d}  {us:8.1f} us  {2.0 * R * P * Q / us / 1e6:7.0f} TF/s")
