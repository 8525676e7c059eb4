"""Developer script (GPU box): duration of attn_fwd_kernel (encoder call: B x 20 heads, 1500 x 1500) for the library WFT_LIB points at
(tools/dev/fwd_abl.sh builds the ablation variants).   for n in "" 1 2 ...; do WFT_LIB=.../libwft_fwdabl$n.so python tools/dev/fwd_abl.py; done"""
import os, sys, torch
from pathlib import Path
R = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
for B in (32, 68):
    H, T = 20, 1500
    d = H * 64
    qkv = torch.randn(B, T, 3 * d, device=dev).bfloat16()
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    for _ in range(3): K.attn_fwd(q, k, v, H, False, 0.125)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(5): K.attn_fwd(q, k, v, H, False, 0.125)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    out.append(f"B={B}: {min(ts):.1f} us ({4.0 * T * T * 64 * H * B / min(ts) / 1e6:.0f} TF/s)")
print(f"{os.path.basename(os.environ.get('WFT_LIB', 'libwft.so')):24s} " + " | ".join(out), flush=True)
