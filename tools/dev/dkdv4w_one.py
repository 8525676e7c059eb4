"""Developer script (GPU box): encoder-shaped attention backward per dK/dV variant under rocprofv3 (kernel durations), at Tq = 1500
and 3000 with Tk = 1500 (slope = cost per query block, intercept = prologue + epilogue).
  rocprofv3 --kernel-trace -d DIR -o x -- python3 tools/dev/dkdv4w_one.py [B];  python3 tools/dev/dkdv4w_one.py parse DIR"""
import sys
from pathlib import Path
if len(sys.argv) > 2 and sys.argv[1] == "parse":
    import csv, glob, collections
    f = sorted(glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "attn" in n:
            acc[(n, r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (n, g), v in sorted(acc.items()):
        v = sorted(v)
        print(f"{n:36s} grid {g:>9s} n {len(v):3d}  median {v[len(v) // 2]:9.1f} us  min {v[0]:9.1f}")
    sys.exit(0)
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, Tk = 20, 1500
import os
for Tq in ((1500,) if os.environ.get('ONE') else (1500, 3000)):
    q = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Tk, 2 * H * 64, device=dev).to(torch.bfloat16)
    k, v = kv[..., :H * 64], kv[..., H * 64:]
    do = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    cs = (torch.empty(H * 64, device=dev), torch.empty(H * 64, device=dev))
    for var in ((1, 0) if os.environ.get('ONE') else (1, 0, 1, 0)):
        K.set_variant("dkdv", var); K.set_variant("dq", var); K.set_variant("fwd", var)
        for _ in range(3):
            K.attn_fwd(q, k, v, H, False, 0.125)
        for _ in range(2 if os.environ.get('ONE') else 5):
            K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125, colsums=cs)
        torch.cuda.synchronize()
