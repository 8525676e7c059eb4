"""Developer script (GPU box): A/B of two libwft.so builds in ONE gpurun call — interleaved rounds, one child process per
library (the path is read once, from WFT_LIB).  Usage:
    python tools/dev/lib_ab.py [--rounds 3] [--what gemm|attn|all] <label>=<path/to/libwft.so> ...
`gemm`: NT256 per-tile fixed cost (M = 65 536, N = 1280: 5 tiles per CU; K = 1280 / 5120 two-point fit) and the encoder-shape
epilogue variants at 68 x 1500 rows.  `attn`: forward / backward at the encoder shape and the decoder's cross shape."""
import json, os, subprocess, sys, time


def child(what):
    import torch
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
    from whisper_finetune.engine import kernels as K, lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    def bench(f, n=20):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3  # us

    out = {}
    if what in ("gemm", "all"):
        M, N = 65536, 1280
        fit = {}
        for Kd in (1280, 5120):
            a = torch.randn(M, Kd, device=dev).to(torch.bfloat16); b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            r = torch.randn(M, N, device=dev).to(torch.bfloat16)
            fit[Kd] = (bench(lambda: K.gemm_nt(a, b, out=o)) / 5, bench(lambda: K.gemm_nt(a, b, out=o, residual=r)) / 5)
        sl = (fit[5120][0] - fit[1280][0]) / 3840; slr = (fit[5120][1] - fit[1280][1]) / 3840
        out["seam_us"] = round(fit[1280][0] - sl * 1280, 2); out["seam_res_us"] = round(fit[1280][1] - slr * 1280, 2)
        out["ns_per_k"] = round(sl * 1e3, 2)
        M = 68 * 1500
        for N, Kd in ((5120, 1280), (1280, 5120), (1280, 1280), (3840, 1280)):
            a = torch.randn(M, Kd, device=dev).bfloat16(); b = (torch.randn(N, Kd, device=dev) * 0.03).bfloat16()
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            aux = torch.randn(M, N, device=dev).bfloat16(); res = torch.randn(M, N, device=dev).bfloat16()
            bias = torch.randn(N, device=dev); cs = torch.empty(N, device=dev)
            cases = {"bias": lambda: K.gemm_nt(a, b, out=o, bias=bias)}
            if N == 5120:
                cases["gelu_grad"] = lambda: K.gemm_nt(a, b, out=o, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)
                cases["mulaux+cs"] = lambda: K.gemm_nt(a, b, out=o, epilogue=L.EPI_MUL_AUX, aux=aux, colsum=cs)
            else:
                cases["bias+res"] = lambda: K.gemm_nt(a, b, out=o, bias=bias, residual=res)
                cases["none+cs"] = lambda: K.gemm_nt(a, b, out=o, colsum=cs)
            for name, fn in cases.items():
                out[f"{N}x{Kd}:{name}"] = round(2.0 * M * N * Kd / bench(fn, 10) / 1e6, 1)  # TF/s
    if what in ("tn", "all"):
        R = 68 * 1500
        for P, Q in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
            a = torch.randn(R, P, device=dev).bfloat16(); b = torch.randn(R, Q, device=dev).bfloat16()
            o = torch.empty(P, Q, dtype=torch.float32, device=dev)
            out[f"tn_{P}x{Q}"] = round(2.0 * R * P * Q / bench(lambda: K.gemm_tn(a, b, out=o), 10) / 1e6, 1)  # TF/s
    if what in ("attn", "all"):
        for tag, (B, H, Tq, Tk, causal) in {"enc": (32, 20, 1500, 1500, False), "cross": (32, 20, 128, 1500, False),
                                            "self": (32, 20, 448, 448, True)}.items():
            d = H * 64
            if Tq == Tk:
                qkv = torch.randn(B, Tq, 3 * d, device=dev).bfloat16()
                q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            else:
                q = torch.randn(B, Tq, d, device=dev).bfloat16()
                kv = torch.randn(B, Tk, 2 * d, device=dev).bfloat16()
                k, v = kv[..., :d], kv[..., d:]
            do = torch.randn(B, Tq, d, device=dev).bfloat16()
            o, lse = K.attn_fwd(q, k, v, H, causal, 0.125)
            out[f"attn_{tag}_fwd_us"] = round(bench(lambda: K.attn_fwd(q, k, v, H, causal, 0.125), 10), 1)
            out[f"attn_{tag}_bwd_us"] = round(bench(lambda: K.attn_bwd(q, k, v, o, lse, do, H, causal, 0.125), 10), 1)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    args = sys.argv[1:]
    rounds, what = 3, "gemm"
    while args and args[0].startswith("--"):
        if args[0] == "--rounds": rounds = int(args[1])
        elif args[0] == "--what": what = args[1]
        args = args[2:]
    libs = [a.split("=", 1) for a in args]  # label=path[:VAR=VALUE[:VAR=VALUE ...]]
    acc = {lab: {} for lab, _ in libs}
    for rnd in range(rounds):
        for lab, spec in libs:
            path, *envs = spec.split(":")
            env = dict(os.environ, WFT_LIB=os.path.abspath(path))
            env.update(dict(e.split("=", 1) for e in envs))
            r = subprocess.run([sys.executable, __file__, "child", what], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(lab, "FAILED", r.stderr[-600:], flush=True)
                continue
            for k, v in json.loads(line[-1][7:]).items(): acc[lab].setdefault(k, []).append(v)
    keys = list(next(iter(acc.values())).keys()) if acc else []
    print(f"{'':28s}" + "".join(f"{lab:>26s}" for lab, _ in libs))
    libs = [(lab, None) for lab, _ in libs]
    for k in keys:
        row = f"{k:28s}"
        for lab, _ in libs:
            v = sorted(acc[lab].get(k, [float('nan')]))
            row += f"{v[len(v) // 2]:14.1f} ({v[0]:.1f}-{v[-1]:.1f})".rjust(26)
        print(row, flush=True)
