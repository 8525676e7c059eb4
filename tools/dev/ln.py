"""dev tool: LayerNorm forward/backward bandwidth at the bench shape (rows = B*1500, cols = 1280)."""
import sys, torch
sys.path.insert(0, "whisper-finetune_amd")
from whisper_finetune.engine import kernels as K

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, cols in ((68, 1280), (32, 1280), (68, 512), (8, 384)):
    rows = B * 1500
    x = torch.randn(rows, cols, device="cuda").bfloat16()
    dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.randn(cols, device="cuda"); b = torch.randn(cols, device="cuda")
    y, mean, rstd = K.layernorm_fwd(x, g, b)
    nb = rows * cols * 2
    us = t(lambda: K.layernorm_fwd(x, g, b))
    print(f"B={B} cols={cols} fwd {us:7.1f} us  {2*nb/us/1e6:5.2f} TB/s")
    us = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_colsum=True))
    print(f"B={B} cols={cols} bwd(colsum,dres) {us:7.1f} us  {4*nb/us/1e6:5.2f} TB/s")
    us = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd))
    print(f"B={B} cols={cols} bwd(plain) {us:7.1f} us  {3*nb/us/1e6:5.2f} TB/s")
