import os, sys, time, subprocess
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__))); import _timing_lib  # noqa: E702 (WFT_LIB -> libwft_timing.so)
# A/B of WFT_GEMM_DIAG variants in interleaved rounds (each variant in its own process: the library reads the variable once)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = []
    for M, N, Kd in ((48000, 5120, 1280), (48000, 1280, 5120), (48000, 3840, 1280), (102000, 1280, 1280)):
        a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for _ in range(5): K.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): K.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res.append(2 * M * N * Kd / dt / 1e12)
    print(" ".join(f"{r:7.1f}" for r in res))
else:
    for rnd in range(3):
        for d in sys.argv[1:]:
            env = dict(os.environ, WFT_GEMM_DIAG=d)
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
            print(f"diag={d}: {out}", flush=True)
