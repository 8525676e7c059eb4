import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import lib as L
    if os.environ.get("WFT_LIB_OVERRIDE"):
        L.LIB_PATH = Path(os.environ["WFT_LIB_OVERRIDE"]).resolve()
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = []
    for R, P, Q in ((102000, 1280, 1280), (102000, 3840, 1280), (102000, 5120, 1280), (102000, 1280, 5120)):
        a = torch.randn(R, P, device=dev).to(torch.bfloat16); b = torch.randn(R, Q, device=dev).to(torch.bfloat16)
        for _ in range(3): K.gemm_tn(a, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): K.gemm_tn(a, b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        res.append(2 * R * P * Q / dt / 1e12)
    print(" ".join(f"{r:7.1f}" for r in res))
else:
    for rnd in range(3):
        for spec in sys.argv[1:]:
            env = dict(os.environ)
            for kv in spec.split(","):
                k, v = kv.split("="); env[k] = v
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
            print(f"{spec}: {out}", flush=True)
