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
    B, H, T = 32, 20, 1500
    qkv = torch.randn(B, T, 3 * H * 64, device=dev).to(torch.bfloat16)
    q, k, v = qkv[..., :1280], qkv[..., 1280:2560], qkv[..., 2560:]
    do = torch.randn(B, T, 1280, device=dev).to(torch.bfloat16)
    def t(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    print(f"fwd {t(lambda: K.attn_fwd(q, k, v, H, False, 0.125)):.3f} ms  bwd {t(lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)):.3f} ms")
else:
    for rnd in range(3):
        for spec in sys.argv[1:]:
            env = dict(os.environ)
            for kv in spec.split(","):
                k, v = kv.split("="); env[k] = v
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
            print(f"{spec}: {out}", flush=True)
