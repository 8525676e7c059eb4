"""Developer script (GPU box): per-phase cycle sums of the dK/dV attention-backward kernel from an instrumented build
(`-DSTAMPS`, see profiles/README.md round 3): WFT_LIB must point at that build; its `wft_dbg_read` returns the sums."""
import ctypes, os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
so = ctypes.CDLL(os.environ["WFT_LIB"])
B, H, T = 32, 20, 1500
d = H * 64
qkv = torch.randn(B, T, 3 * d, device=dev).bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
do = torch.randn(B, T, d, device=dev).bfloat16()
o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
for _ in range(3): K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
so.wft_dbg_read(buf, 1)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
e1.record(); torch.cuda.synchronize()
so.wft_dbg_read(buf, 0)
w = buf[8]
names = ["barrier exit -> tile start", "stage issue", "LDS reads issued -> landed", "S / dP MFMA issue", "MFMA tail + exp + pack (+ 2nd wait)",
         "dV / dK MFMA issue", "tile end -> barrier exit"]
print(f"bwd {e0.elapsed_time(e1) / n * 1e3:.1f} us per call; {w} waves sampled; kernel lifetime per workgroup {buf[9] / w:.0f} clock ticks")
halves = (T + 31) // 32
for i, nm in enumerate(names):
    print(f"  {nm:40s} {buf[i] / w:10.0f} ticks per workgroup   {buf[i] / w / halves:8.1f} per 32-query half   {100.0 * buf[i] / buf[9]:5.1f} %")
