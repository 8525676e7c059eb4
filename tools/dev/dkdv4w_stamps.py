"""Developer script (GPU box): where a workgroup of attn_bwd_dkdv4w_kernel spends its clock ticks (build with -DD4_STAMPS, see
csrc/attn.hip; WFT_LIB points at that build).   WFT_LIB=.../libwft_stamps.so python tools/dev/dkdv4w_stamps.py"""
import ctypes, os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
lib = L.load()
so = ctypes.CDLL(os.environ["WFT_LIB"])
B, H, T = 32, 20, 1500
TQ = int(sys.argv[1]) if len(sys.argv) > 1 else T  # (128: the cross-attention call, Tq = 128 queries over 1 500 keys)
if len(sys.argv) > 2:
    B = int(sys.argv[2])
d = H * 64
qkv = torch.randn(B, T, 3 * d, device=dev).bfloat16()
q, k, v = qkv[:, :TQ, :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
do = torch.randn(B, TQ, d, device=dev).bfloat16()
o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
cs = (torch.empty(d, device=dev), torch.empty(d, device=dev))
for _ in range(3): K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
so.wft_dbg_read(buf, 1)
K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
so.wft_dbg_read(buf, 0)
names = ("prologue (descriptors, requests or wait, zeroing, barrier)", "block 0 + loop", "last dV / dK + drain", "prefetch block",
         "read-back + colsum + stores")
for o, what in ((0, "first item of a workgroup"), (8, "later items")):
    n = buf[o]
    if not n:
        continue
    print(f"{what}: {n} items; clock ticks per item")
    for i, name in enumerate(names):
        print(f"  {name:60s} {buf[o + 1 + i] / n:10.0f}")
    print(f"  sum {sum(buf[o + 1:o + 6]) / n:.0f}")
