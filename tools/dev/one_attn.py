import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, T = 32, 20, 1500
qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(torch.bfloat16)
q, k, v = qkv[..., :1280], qkv[..., 1280:2560], qkv[..., 2560:]
do = torch.randn(B, T, 1280, device=dev).to(torch.bfloat16)
for _ in range(3):
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
torch.cuda.synchronize()
