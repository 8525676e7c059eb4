"""Developer script (GPU box, libwft_timing.so): the decoder-sized weight-gradient GEMMs (R = B x S = 8704 / 11136 rows, 1280 x 1280 and
1280 x 5120 outputs) through the 128-tile kernel (25 output tiles of 256 x 256 < WFT_TN256_MIN_OUT_TILES = 50) and through the 4-wave
256 x 256 kernel (WFT_TN256_MIN_OUT_TILES=1), one process per setting:
    for t in 50 1; do WFT_TN256_MIN_OUT_TILES=$t python tools/dev/tn_small.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _timing_lib  # noqa: E702
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
out = []
for R in (8704, 11136, 4096):
    for P, Q in ((1280, 1280), (2560, 1280), (1280, 5120), (5120, 1280), (3840, 1280)):
        dy = torch.randn(R, P, device=dev).bfloat16(); x = torch.randn(R, Q, device=dev).bfloat16()
        us = t(lambda: K.gemm_tn(dy, x))
        out.append(f"R={R} {P}x{Q}: {us:.1f} us {2*R*P*Q/us/1e6:.0f} TF/s")
print(f"MIN_OUT_TILES={os.environ.get('WFT_TN256_MIN_OUT_TILES', '50')}: " + " | ".join(out), flush=True)
