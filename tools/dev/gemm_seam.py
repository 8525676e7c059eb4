"""Developer script (GPU box): fixed cost per 256x256 tile of gemm_nt256_kernel — time vs K at fixed M, N (tiles per CU fixed),
for the store-only epilogue, and with WFT_GEMM_DIAG=6 (staged epilogue skipped: timing only) in a child process."""
import os, sys, time, subprocess
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__))); import _timing_lib  # noqa: E702 (WFT_LIB -> libwft_timing.so)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    def bench(f, n=20):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    M, N = 65536, 1280  # 256 x 5 = 1280 tiles = exactly 5 per CU
    res = []
    for Kd in (256, 512, 1024, 1280, 2560, 5120):
        a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16)
        t = bench(lambda: K.gemm_nt(a, b, out=out)) / 5
        tr = bench(lambda: K.gemm_nt(a, b, out=out, residual=r)) / 5
        res.append((Kd, t, tr))
    print(" ".join(f"K={k}: {t:.1f}/{tr:.1f}us" for k, t, tr in res))
    (k0, t0, r0), (k1, t1, r1) = res[3], res[5]
    b_ = (t1 - t0) / (k1 - k0); a_ = t0 - b_ * k0
    br = (r1 - r0) / (k1 - k0); ar = r0 - br * k0
    print(f"per tile: plain {a_:.2f} us + {b_*1e3:.2f} ns*K | +residual {ar:.2f} us + {br*1e3:.2f} ns*K")
else:
    for spec in ("WFT_X=0", "WFT_GEMM_DIAG=6", "WFT_GEMM_DIAG=8", "WFT_NT256_PERSISTENT=0"):
        env = dict(os.environ); k, v = spec.split("="); env[k] = v
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(spec); print("\n".join(r.stdout.strip().splitlines()[-2:]) or r.stderr[-400:], flush=True)
