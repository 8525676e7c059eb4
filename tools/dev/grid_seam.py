"""Developer script (GPU box): is the NT256 tile seam a chip-wide burst effect?  The same 5-tiles-per-workgroup problem on
persistent grids of 32 ... 256 workgroups (WFT_NT256_GRID; M scales with the grid), K = 1280 and 5120: per-tile time and the
two-point fixed cost.  A seam that shrinks with the grid is contention for HBM / fabric (all CUs store their tiles at once)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    grid = int(os.environ["WFT_NT256_GRID"])
    M, N = grid * 256, 1280
    def bench(f, n=20):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = {}
    for Kd in (1280, 5120):
        a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev); r = torch.randn(M, N, device=dev).bfloat16()
        t[Kd] = (bench(lambda: K.gemm_nt(a, b, out=o)) / 5, bench(lambda: K.gemm_nt(a, b, out=o, residual=r)) / 5)
    sl = (t[5120][0] - t[1280][0]) / 3840; slr = (t[5120][1] - t[1280][1]) / 3840
    print(f"grid {grid:3d}: tile K=1280 {t[1280][0]:.1f} us (+res {t[1280][1]:.1f}), K=5120 {t[5120][0]:.1f} (+res {t[5120][1]:.1f}); "
          f"fixed {t[1280][0] - sl * 1280:.1f} us + {sl * 1e3:.1f} ns*K | +res fixed {t[1280][1] - slr * 1280:.1f} us + {slr * 1e3:.1f} ns*K", flush=True)
else:
    for g in (32, 64, 128, 192, 256):
        env = dict(os.environ, WFT_NT256_GRID=str(g))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-400:], flush=True)
