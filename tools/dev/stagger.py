"""Developer script (GPU box): NT256 start stagger sweep (WFT_NT256_STAGGER ns per phase x WFT_NT256_PHASES), 5 tiles per CU."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    def bench(f, n=20):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    outs = []
    for (M, N, Kd) in ((65536, 1280, 1280), (65536, 1280, 5120), (102000, 1280, 1280), (102000, 3840, 1280), (102000, 5120, 1280), (102000, 1280, 5120)):
        a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev); r = torch.randn(M, N, device=dev).bfloat16()
        outs.append(f"{bench(lambda: K.gemm_nt(a, b, out=o)):7.1f}/{bench(lambda: K.gemm_nt(a, b, out=o, residual=r)):7.1f}")
    print(" ".join(outs), flush=True)
else:
    print("us plain/+res: 65536x1280x1280 65536x1280x5120 102000x1280x1280 102000x3840x1280 102000x5120x1280 102000x1280x5120")
    for rnd in range(1):
        for st, ph in ((0, 1), (4000, 2), (8000, 2), (2000, 4), (4000, 4), (8000, 4), (2000, 8), (4000, 8)):
            env = dict(os.environ, WFT_NT256_STAGGER=str(st), WFT_NT256_PHASES=str(ph))
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"stagger {st:5d} ns x {ph} phases: " + (r.stdout.strip() or r.stderr[-400:]), flush=True)
