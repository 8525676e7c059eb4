"""Developer script (GPU box): stress the split-K weight-gradient GEMMs for run-to-run bit differences."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for R, P, Q, pv in ((24000, 1280, 128, 0), (48000, 128, 1280, 0), (48000, 128, 1280, 16), (3000, 384, 1536, 0), (24000, 1280, 1280, 0)):
    g = torch.Generator(device=dev).manual_seed(R + P)
    a = torch.randn(R, P, device=dev, generator=g).to(torch.bfloat16); b = torch.randn(R, Q, device=dev, generator=g).to(torch.bfloat16)
    if pv:
        a[:, pv:] = 0
    first = K.gemm_tn(a, b, p_valid=pv).clone()
    bad = 0; worst = 0.0
    for i in range(n):
        c = K.gemm_tn(a, b, p_valid=pv)
        if not torch.equal(c, first):
            bad += 1
            d = (c - first).abs()
            worst = max(worst, d.max().item())
            if bad <= 3:
                idx = (d > 0).nonzero()
                print(f"   run {i}: {idx.shape[0]} elements differ, rows {idx[:,0].min().item()}..{idx[:,0].max().item()} cols {idx[:,1].min().item()}..{idx[:,1].max().item()} max {d.max().item():.3e}")
    print(f"R={R} P={P} Q={Q} p_valid={pv}: {bad}/{n} runs differ (max |diff| {worst:.3e})", flush=True)
