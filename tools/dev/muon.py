import sys, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
from oracle import whisper_oracle as O
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(1)
for shape in [(128, 384), (384, 128), (256, 256), (16, 320), (320, 16), (1280, 1280)]:
    g = torch.randn(shape) * 0.02
    p = torch.zeros(shape, device=dev)
    buf = torch.zeros(shape, device=dev)
    upd = K.muon_group_step([p], [g.clone().to(dev)], [buf], 1.0, 0.0, 0.95, return_update=True)[0].cpu()
    mom = torch.zeros(shape)
    ref = O.muon_update(g.clone(), mom, beta=0.95).float()
    rel = ((upd - ref).norm() / ref.norm()).item()
    step_rel = ((-p.cpu() - ref).norm() / ref.norm()).item()
    print(shape, "update rel", rel, "applied rel", step_rel, "buf err", (buf.cpu() - mom).abs().max().item(),
          "sv", torch.linalg.svdvals(upd)[[0, -1]].tolist(), "ref sv", torch.linalg.svdvals(ref)[[0, -1]].tolist())
# timing: large-v3 full-FT shapes
import time
for n, shape in [(384, (1280, 1280)), (64, (5120, 1280)), (64, (1280, 5120)), (512, (16, 1280)), (512, (1280, 16))]:
    ps = [torch.zeros(shape, device=dev) for _ in range(n)]
    gs = [torch.randn(shape, device=dev) for _ in range(n)]
    bs = [torch.zeros(shape, device=dev) for _ in range(n)]
    K.muon_group_step(ps, gs, bs, 1e-3, 0.0, 0.95)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K.muon_group_step(ps, gs, bs, 1e-3, 0.0, 0.95)
    torch.cuda.synchronize(); print(n, shape, f"{(time.perf_counter() - t0) * 1e3:.2f} ms")
