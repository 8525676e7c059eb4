"""Developer script (GPU box): cross-attention backward (Tq = 128, Tk = 1500) on the one-wave-per-SIMD dK/dV kernel vs the 8-wave kernel."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0"); torch.manual_seed(0)
for B, Tq in ((96, 128), (32, 128), (32, 448), (96, 64)):
    H, Tk = 20, 1500; d = H * 64
    q = torch.randn(B, Tq, d, device=dev).bfloat16(); kv = torch.randn(B, Tk, 2 * d, device=dev).bfloat16()
    k, v = kv[..., :d], kv[..., d:]
    do = torch.randn(B, Tq, d, device=dev).bfloat16()
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    cs = (torch.empty(d, device=dev), torch.empty(d, device=dev))
    res = {}
    for var in (0, 1):
        K.set_variant("dkdv", var)
        f = lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(4): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        res[var] = min(ts)
    K.set_variant("dkdv", 0)
    print(f"B={B} {Tq}x{Tk}: dQ 8-wave + dK/dV 4-wave {res[0]:.1f} us | dQ 8-wave + dK/dV 8-wave {res[1]:.1f} us", flush=True)
