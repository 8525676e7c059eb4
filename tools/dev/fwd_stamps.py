"""Developer script (GPU box): where a wave of attn_fwd_kernel spends its clock ticks per 64-key tile (encoder call: 20 heads, 1500 x 1500;
build: tools/dev/fwd_stamps.sh -> libwft_fwdstamps.so, selected through WFT_LIB).  Ideal per tile and wave: 16 MFMA 32x32x16 = 512
matrix-pipe cycles; the vector stream is 32 v_exp_f32 (8 issue cycles each) + ~100 other vector instructions (4.5 each)."""
import ctypes, os, sys, torch
from pathlib import Path
R = Path(__file__).resolve().parents[2]
os.environ.setdefault("WFT_LIB", str(R / "whisper-finetune_amd" / "libwft_fwdstamps.so"))
sys.path.insert(0, str(R / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
L.load()
so = ctypes.CDLL(os.environ["WFT_LIB"])
B, H, T = int(os.environ.get("FWD_B", 32)), 20, 1500
d = H * 64
qkv = torch.randn(B, T, 3 * d, device=dev).bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
for _ in range(3): K.attn_fwd(q, k, v, H, False, 0.125)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
so.wft_fwd_dbg_read(buf, 1)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): K.attn_fwd(q, k, v, H, False, 0.125)
e1.record(); torch.cuda.synchronize()
so.wft_fwd_dbg_read(buf, 0)
w, tiles = buf[8], buf[10]
names = ["tile entry -> stage issue done", "V^T read issue + S MFMA chains complete", "maximum (tree + half exchange)",
         "(rescale,) exponentials, sums, packs", "s_waitcnt: own LDS-DMA of tile kt+1 + V^T landed", "barrier",
         "K reads of kt+1 issued + P.V MFMA chains complete"]
print(f"fwd {e0.elapsed_time(e1) / n * 1e3:.1f} us per call (instrumented); {w} waves sampled, {tiles / w:.1f} tiles per wave; "
      f"wave lifetime {buf[9] / w:.0f} ticks = {buf[9] / tiles:.0f} per tile")
for i, nm in enumerate(names):
    print(f"  {nm:55s} {buf[i] / tiles:8.1f} ticks per tile   {100.0 * buf[i] / buf[9]:5.1f} %")
print(f"  {'prologue + epilogue (outside the tiles)':55s} {(buf[9] - sum(buf[:7])) / w:8.1f} ticks per wave")
