"""Developer script (GPU box): attention backward A/B — child processes with different WFT_ATTN_* settings, encoder shape
(B x 20 heads x 1500 x 1500), decoder causal and cross shapes.   python tools/dev/attn_bwd_ab.py WFT_ATTN_DKDV=32 WFT_ATTN_DKDV=64"""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    def t(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    out = []
    for B, H, Tq, Tk, causal in ((32, 20, 1500, 1500, False), (68, 20, 128, 1500, False)):
        q = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
        kv = torch.randn(B, Tk, 2 * H * 64, device=dev).to(torch.bfloat16)
        k, v = kv[..., :H * 64], kv[..., H * 64:]
        do = torch.randn(B, Tq, H * 64, device=dev).to(torch.bfloat16)
        o, lse = K.attn_fwd(q, k, v, H, causal, 0.125)
        msf = t(lambda: K.attn_fwd(q, k, v, H, causal, 0.125))
        ms = t(lambda: K.attn_bwd(q, k, v, o, lse, do, H, causal, 0.125))
        fl = 10.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)  # 5 algorithmic products
        out.append(f"{Tq}x{Tk}{'c' if causal else ''}: fwd {msf:.3f} bwd {ms:.3f} ms ({fl / ms / 1e9:.0f} TF alg)")
    print(" | ".join(out))
else:
    for rnd in range(2):
        for spec in sys.argv[1:]:
            env = dict(os.environ)
            for kv in spec.split(","):
                k, v = kv.split("="); env[k] = v
            res = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"{spec}: {(res.stdout.strip().splitlines() or [res.stderr[-300:]])[-1]}", flush=True)
