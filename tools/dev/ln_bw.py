"""Developer script (GPU box): achieved HBM rate of LayerNorm forward / backward at the headline shape (rows = B x 1500, d = 1280).
The backward moves FOUR tensors when the residual gradient is fused (dy, x, dres in; dx out) — every block LayerNorm of the model
— and three without; the round-4 review priced ln_bwd_kernel<true,5,2> (187.6 us) against three."""
import sys, time
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
for B in (68, 87, 32):
    rows, d = B * 1500, 1280
    x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn(rows, d, device=dev).bfloat16(); dres = torch.randn(rows, d, device=dev).bfloat16()
    g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, g, b)
    nbytes = rows * d * 2
    tf = t(lambda: K.layernorm_fwd(x, g, b))
    t3 = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, want_colsum=True))
    t4 = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_colsum=True))
    t4n = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_colsum=False, want_params=False))
    print(f"B={B}: fwd {tf:.1f} us = {2*nbytes/tf/1e6:.2f} TB/s | bwd 3 tensors {t3:.1f} us = {3*nbytes/t3/1e6:.2f} TB/s | bwd 4 tensors (dres) {t4:.1f} us = {4*nbytes/t4/1e6:.2f} TB/s"
          f" | 4 tensors, dx only {t4n:.1f} us = {4*nbytes/t4n/1e6:.2f} TB/s  (times include the two reduce launches and the wrapper's allocations)", flush=True)
