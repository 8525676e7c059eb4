"""Developer script (GPU box): LayerNorm backward calls of a small model, timed alone (kernel + its reduce launches)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
for rows, cols in [(1024, 512), (12000, 512), (1024, 1280), (16000, 768), (144000, 1280)]:
    x = torch.randn(rows, cols, device=dev).bfloat16(); dy = torch.randn(rows, cols, device=dev).bfloat16()
    gamma = torch.randn(cols, device=dev); beta = torch.randn(cols, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta)
    fn = lambda: K.layernorm_bwd(dy, x, gamma, mean, rstd)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"ln_bwd {rows:7d} {cols:5d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
for rows, cols in [(24000, 512), (12000, 512), (1024, 2048)]:
    x = torch.randn(rows, cols, device=dev).bfloat16()
    fn = lambda: K.colsum(x)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"colsum {rows:7d} {cols:5d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
