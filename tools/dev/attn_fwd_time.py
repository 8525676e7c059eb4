"""Developer script (GPU box): time of the encoder attention forward for the library WFT_LIB points at."""
import os, sys, torch
from pathlib import Path
R = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0"); torch.manual_seed(0)
out = []
for B, Tq, Tk in ((32, 1500, 1500), (87, 1500, 1500)):
    H = 20; d = H * 64
    q = torch.randn(B, Tq, d, device=dev).bfloat16(); kv = torch.randn(B, Tk, 2 * d, device=dev).bfloat16()
    k, v = kv[..., :d], kv[..., d:]
    PRE = os.environ.get("WFT_TIME_PRE", "0") == "1"
    f = lambda: K.attn_fwd(q, k, v, H, False, 0.125, q_prescaled=PRE)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record()
        for _ in range(4): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    out.append(f"B={B} {Tq}x{Tk}: {min(ts):.1f} us")
print(("pre " if os.environ.get("WFT_TIME_PRE", "0") == "1" else "    ") + f"fwd {os.path.basename(os.environ.get('WFT_LIB', 'libwft.so')):22s} " + " | ".join(out), flush=True)
