"""Developer script (GPU box): per-tile fixed cost ("seam") of the 256x256 NT kernels from a K sweep at M = 65 536 (exactly
tiles/256 tiles per CU): time per tile = S + T * (K / 64).  python tools/dev/nt4w_seam.py [N]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0"); lib = L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
M = 65536
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(torch.bfloat16)
for ename, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+res", dict(bias=bias, residual=res)), ("mul_aux", dict(epilogue=L.EPI_MUL_AUX, aux=res)), ("gelu_grad", dict(bias=bias, epilogue=L.EPI_GELU_GRAD, aux=res))):
    for v, name in ((1, "pp"), (0, "4w")):
        K.set_variant("nt", v)
        pts = []
        for Kd in (256, 512, 1280, 2560, 5120):
            a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = (torch.randn(N, Kd, device=dev) * 0.03).to(torch.bfloat16)
            t = min(timeit(lambda: K.gemm_nt(a, b, out=out, **kw)) for _ in range(3))
            per_tile = t / (M // 256 * (N // 256) / 256) * 1e6
            pts.append((Kd // 64, per_tile))
        # least squares fit
        n = len(pts); sx = sum(x for x, _ in pts); sy = sum(y for _, y in pts); sxx = sum(x * x for x, _ in pts); sxy = sum(x * y for x, y in pts)
        T = (n * sxy - sx * sy) / (n * sxx - sx * sx); S = (sy - T * sx) / n
        print(f"N={N} {ename:10s} {name}: seam {S:6.2f} us  k-step {T:6.3f} us   tiles: " + " ".join(f"{y:.1f}" for _, y in pts), flush=True)
