"""Developer script (not collected by pytest): exercises every libwft kernel on the GPU
against torch fp32 math and prints error / timing tables.  Run on the GPU box:
    python tools/dev/kernels.py [--perf]
"""
import math
import sys
import time
import traceback
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
results = []


def report(name, got, ref, tol):
    got = got.float()
    ref = ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    rel = err / scale
    ok = rel <= tol and math.isfinite(err)
    results.append((name, ok))
    print(f"{'PASS' if ok else 'FAIL'} {name:50s} max_abs={err:.3e} rel_to_max={rel:.3e} tol={tol:.1e}", flush=True)
    return ok


def run(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()
        results.append((fn.__name__, False))
        print(f"FAIL {fn.__name__} (exception)", flush=True)


def bf(x):
    return x.to(torch.bfloat16)


def t_cast():
    x = torch.randn(1000, 77, device=dev)
    report("cast_f32_bf16", K.cast_bf16(x), bf(x), 0)
    w = torch.randn(300, 200, device=dev)
    d, dt = K.weight_shadow(w, 384, 256, True)
    ref = torch.zeros(384, 256, device=dev)
    ref[:300, :200] = w
    report("weight_shadow", d, bf(ref), 0)
    report("weight_shadow_T", dt, bf(ref).t(), 0)
    a, b = bf(torch.randn(4097, device=dev)), bf(torch.randn(4097, device=dev))
    report("add_bf16", K.add_bf16(a, b), bf(a.float() + b.float()), 0)
    x = bf(torch.randn(5000, 384, device=dev))
    report("colsum", K.colsum(x), x.float().sum(0), 2e-3)


def t_ln():
    for rows, cols in ((3000, 384), (777, 1280), (100, 512)):
        x = bf(torch.randn(rows, cols, device=dev) * 2 + 0.5)
        g = torch.randn(cols, device=dev)
        b = torch.randn(cols, device=dev)
        y, mean, rstd = K.layernorm_fwd(x, g, b)
        ref = torch.nn.functional.layer_norm(x.float(), (cols,), g, b)
        report(f"ln_fwd {rows}x{cols}", y, ref, 1e-2)
        dy = bf(torch.randn(rows, cols, device=dev))
        dres = bf(torch.randn(rows, cols, device=dev))
        xr = x.float().requires_grad_(True)
        gr = g.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (cols,), gr, br).backward(dy.float())
        dx, dg, db = K.layernorm_bwd(dy, x, g, mean, rstd, dres)
        report(f"ln_bwd dx {rows}x{cols}", dx, xr.grad + dres.float(), 1e-2)
        report(f"ln_bwd dgamma {rows}x{cols}", dg, gr.grad, 2e-3)
        report(f"ln_bwd dbeta {rows}x{cols}", db, br.grad, 2e-3)
    # masked variant (deep SpecAugment)
    B, T, cols = 2, 150, 384
    x = bf(torch.randn(B * T, cols, device=dev))
    g = torch.randn(cols, device=dev)
    b = torch.randn(cols, device=dev)
    mask = (T, 10, 35, 100, 120)
    y, mean, rstd = K.layernorm_fwd(x, g, b, mask=mask)
    ref = torch.nn.functional.layer_norm(x.float(), (cols,), g, b).view(B, T, cols).clone()
    ref[:, 10:35, :] = 0
    ref[:, :, 100:120] = 0
    report("ln_fwd masked", y, ref.view(B * T, cols), 1e-2)


def t_gemm_nt():
    for (M, N, K_) in ((256, 128, 64), (300, 256, 128), (1500, 384, 384), (4096, 1280, 1280), (777, 1536, 5120)):
        a = bf(torch.randn(M, K_, device=dev))
        b = bf(torch.randn(N, K_, device=dev))
        bias = torch.randn(N, device=dev)
        res = bf(torch.randn(M, N, device=dev))
        ref = a.float() @ b.float().t()
        report(f"gemm_nt {M}x{N}x{K_}", K.gemm_nt(a, b), ref, 1e-2)
        report(f"gemm_nt f32out {M}x{N}x{K_}", K.gemm_nt(a, b, out_f32=True), ref, 1e-5)
        report(f"gemm_nt bias+res {M}x{N}x{K_}", K.gemm_nt(a, b, bias=bias, residual=res), ref + bias + res.float(), 1e-2)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out = K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU, aux=aux)
        report(f"gemm_nt gelu {M}x{N}x{K_}", out, torch.nn.functional.gelu(ref + bias), 1e-2)
        report(f"gemm_nt gelu-pre {M}x{N}x{K_}", aux, ref + bias, 1e-2)
        pre = aux.float().requires_grad_(True)
        torch.nn.functional.gelu(pre).backward(torch.ones_like(pre))
        out = K.gemm_nt(a, b, epilogue=L.EPI_DGELU, aux=aux)
        report(f"gemm_nt dgelu {M}x{N}x{K_}", out, ref * pre.grad, 1e-2)
    # accumulate into f32
    a = bf(torch.randn(256, 128, device=dev)); b = bf(torch.randn(128, 128, device=dev))
    c = torch.ones(256, 128, device=dev)
    K.gemm_nt(a, b, out=c, accumulate=True, alpha=0.5)
    report("gemm_nt accumulate alpha", c, 1 + 0.5 * (a.float() @ b.float().t()), 1e-5)
    # conv-as-gemm: overlapping rows (lda < K), batched, pad-row zeroing
    B, T, Cin, Cout = 2, 100, 128, 256
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, 3, device=dev)
    bias = torch.randn(Cout, device=dev)
    xt = torch.zeros(B, T + 2, Cin, device=dev)
    xt[:, 1:T + 1] = x.transpose(1, 2)
    xt = bf(xt)
    wk = bf(w.permute(0, 2, 1).reshape(Cout, 3 * Cin).contiguous())
    ref = torch.nn.functional.conv1d(bf(x).float(), bf(w).float(), bias, padding=1).transpose(1, 2)  # [B,T,Cout]
    out = torch.full((B, T + 2, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    K.gemm_nt(xt, wk, M=T, N=Cout, K=3 * Cin, lda=Cin, ldb=3 * Cin, out=out[:, 1:], ldc=Cout, bias=bias, batch=B,
              strideA=(T + 2) * Cin, strideC=(T + 2) * Cout)
    report("conv1 s1 as gemm", out[:, 1:T + 1], ref, 1e-2)
    # stride 2
    T2 = T // 2
    ref2 = torch.nn.functional.conv1d(bf(x).float(), bf(w).float(), bias, padding=1, stride=2).transpose(1, 2)
    out2 = K.gemm_nt(xt, wk, M=T2, N=Cout, K=3 * Cin, lda=2 * Cin, ldb=3 * Cin, bias=bias, batch=B,
                     strideA=(T + 2) * Cin, strideC=T2 * Cout)
    report("conv2 s2 as gemm", out2.view(B, T2, Cout), ref2, 1e-2)


def t_gemm_tn():
    for (R, P, Q) in ((64, 128, 128), (100, 128, 256), (1500, 384, 384), (4097, 1280, 256), (333, 256, 1280)):
        a = bf(torch.randn(R, P, device=dev))
        b = bf(torch.randn(R, Q, device=dev))
        ref = a.float().t() @ b.float()
        report(f"gemm_tn {R}:{P}x{Q}", K.gemm_tn(a, b), ref, 1e-5)
        c = torch.ones(P, Q, device=dev)
        K.gemm_tn(a, b, out=c, accumulate=True)
        report(f"gemm_tn acc {R}:{P}x{Q}", c, ref + 1, 1e-5)
    # strided A (column slice of a wider matrix), batch reduction
    big = bf(torch.randn(3, 200, 512, device=dev))
    xb = bf(torch.randn(3, 200, 128, device=dev))
    a = big[:, :, 128:256]
    ref = sum(a[i].float().t() @ xb[i].float() for i in range(3))
    out = K.gemm_tn(a, xb, R=200, P=128, Q=128, lda=512, ldb=128, batch=3, strideA=200 * 512, strideB=200 * 128)
    report("gemm_tn batched strided", out, ref, 1e-5)


def ref_attn(q, k, v, H, causal, scale):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    qh = q.float().view(B, Tq, H, 64).transpose(1, 2)
    kh = k.float().view(B, Tk, H, 64).transpose(1, 2)
    vh = v.float().view(B, Tk, H, 64).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s + torch.full((Tq, Tk), float("-inf"), device=q.device).triu_(1)
    p = torch.softmax(s, -1)
    o = (p @ vh).transpose(1, 2).reshape(B, Tq, D)
    return o, torch.logsumexp(s, -1)


def t_attn():
    for (B, H, Tq, Tk, causal) in ((2, 6, 1500, 1500, False), (2, 4, 128, 128, True), (3, 2, 37, 37, True),
                                   (2, 6, 50, 1500, False), (1, 20, 448, 448, True), (2, 2, 200, 77, False)):
        D = H * 64
        qkv = bf(torch.randn(B, Tq, 3 * D, device=dev))
        if Tq == Tk:
            q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        else:
            q = bf(torch.randn(B, Tq, D, device=dev))
            kv = bf(torch.randn(B, Tk, 2 * D, device=dev))
            k, v = kv[..., :D], kv[..., D:]
        scale = 0.125
        o, lse = K.attn_fwd(q, k, v, H, causal, scale)
        qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, k, v))
        oref, lref = ref_attn(qr, kr, vr, H, causal, scale)
        tag = f"B{B} H{H} Tq{Tq} Tk{Tk} c{int(causal)}"
        report(f"attn_fwd o {tag}", o, oref, 2e-2)
        report(f"attn_fwd lse {tag}", lse, lref, 1e-3)
        do = bf(torch.randn(B, Tq, D, device=dev))
        oref.backward(do.float())
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, H, causal, scale)
        report(f"attn_bwd dq {tag}", dq, qr.grad, 2e-2)
        report(f"attn_bwd dk {tag}", dk, kr.grad, 2e-2)
        report(f"attn_bwd dv {tag}", dv, vr.grad, 2e-2)


def t_embed_ce():
    B, S, d, V = 3, 17, 384, 1000
    tok = torch.randint(0, V, (B, S), device=dev)
    emb = torch.randn(V, d, device=dev)
    pos = torch.randn(448, d, device=dev)
    out = K.embed_fwd(tok, emb, pos)
    report("embed_fwd", out, emb[tok] + pos[:S], 1e-2)
    dout = bf(torch.randn(B, S, d, device=dev))
    demb = torch.zeros(V, d, device=dev); dpos = torch.zeros(448, d, device=dev)
    K.embed_bwd(tok, dout, demb, dpos)
    ref_e = torch.zeros(V, d, device=dev).index_add_(0, tok.view(-1), dout.float().view(-1, d))
    report("embed_bwd demb", demb, ref_e, 1e-5)
    report("embed_bwd dpos", dpos[:S], dout.float().sum(0), 1e-5)
    for V, eps in ((51865, 0.0), (51866, 0.1), (1000, 0.05)):
        rows = 50
        ld = K.round_up(V, 128)
        logits = bf(torch.randn(rows, ld, device=dev) * 3)
        tgt = torch.randint(0, V, (rows,), device=dev)
        tgt[::7] = -100
        lr = logits[:, :V].float().detach().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(lr, tgt, label_smoothing=eps)
        (ref * 0.25).backward()
        row_loss, row_lse, stats, am = K.ce_fwd(logits, tgt, V, eps, want_argmax=True)
        loss = stats[0] / stats[1]
        report(f"ce_fwd V{V} eps{eps}", loss.reshape(1), ref.reshape(1), 1e-5)
        report(f"ce argmax V{V}", am.float(), lr.argmax(-1).float(), 0)
        g = torch.tensor([0.25], device=dev)
        dl = K.ce_bwd(logits.clone(), tgt, V, eps, row_lse, stats, g)
        report(f"ce_bwd V{V} eps{eps}", dl[:, :V], lr.grad, 1e-2)
        report(f"ce_bwd pad V{V}", dl[:, V:], torch.zeros_like(dl[:, V:]), 0)


def t_audio():
    import numpy as np
    from transformers.audio_utils import mel_filter_bank

    for n_mels in (80, 128):
        filt = torch.from_numpy(
            mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T.astype(np.float32)
        ).contiguous().to(dev)
        g = torch.Generator().manual_seed(1234)
        audio = (torch.randn(2, 480000, generator=g) * 0.1).to(dev)
        audio[1, 240000:] = 0  # padded clip
        out = K.logmel(audio, filt)
        win = torch.hann_window(400, device=dev)
        refs = []
        for i in range(2):
            st = torch.stft(audio[i], 400, 160, window=win, return_complex=True)
            mag = st[..., :-1].abs() ** 2
            ms = filt @ mag
            ls = torch.clamp(ms, min=1e-10).log10()
            ls = torch.maximum(ls, ls.max() - 8.0)
            refs.append((ls + 4.0) / 4.0)
        report(f"logmel n_mels={n_mels}", out, torch.stack(refs), 2e-3)
        # specaug: compare against torch grid_sample formulation
        mel = out
        params = torch.tensor([[1, 1500, 40, 100, 180, 10, 30, 0], [0, 0, 0, 2990, 3000, 0, 0, 0]], dtype=torch.int32, device=dev)
        ext = torch.tensor([[3, 5], [0, 0]], dtype=torch.int32, device=dev)
        aug = K.specaug(mel, params, ext)
        ref0 = ref_specaug(mel[0], 1500, 40, 100, 180, 10, 30, 3, 5)
        ref1 = mel[1].clone(); ref1[:, 2990:3000] = 0
        report(f"specaug warp n_mels={n_mels}", aug[0], ref0, 1e-5)
        report(f"specaug nowarp n_mels={n_mels}", aug[1], ref1, 0)
        tm = K.mel_to_tmajor(mel, 128)
        ref_t = torch.zeros(2, 3002, 128, device=dev)
        ref_t[:, 1:3001, :n_mels] = mel.transpose(1, 2)
        report(f"mel_to_tmajor n_mels={n_mels}", tm, bf(ref_t), 0)


def ref_specaug(mel, wp, wd, t0, t1, f0, f1, lo, hi):
    n_mels, Lx = mel.shape
    d = mel.device
    x = torch.tensor([0.0, wp, Lx - 1.0], device=d)
    y = torch.tensor([-1.0, (wp - wd) * 2 / (Lx - 1.0) - 1.0, 1.0], device=d)
    m = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
    m = torch.cat([m[[0]], (m[1:] + m[:-1]) / 2, m[[-1]]])
    xs = torch.linspace(0, Lx - 1, Lx, device=d)
    idx = torch.searchsorted(x[1:].contiguous(), xs)
    dx = x[idx + 1] - x[idx]
    t = (xs - x[idx]) / dx
    tt = t.unsqueeze(0) ** torch.arange(4, device=d).view(-1, 1)
    A = torch.tensor([[1, 0, -3, 2], [0, 1, -2, 1], [0, 0, 3, -2], [0, 0, -1, 1]], dtype=torch.float32, device=d)
    hh = A @ tt
    ys = hh[0] * y[idx] + hh[1] * m[idx] * dx + hh[2] * y[idx + 1] + hh[3] * m[idx + 1] * dx
    grid = torch.stack([ys.view(1, -1).expand(n_mels, -1), torch.linspace(-1, 1, n_mels, device=d).view(-1, 1).expand(-1, Lx)], -1)
    out = torch.nn.functional.grid_sample(mel[None, None], grid[None], align_corners=True)[0, 0]
    out[:, t0:t1] = 0
    out[f0:f1] = 0
    out[:lo] = 0
    out[n_mels - hi:] = 0
    return out


def t_adamw():
    n = 100003
    p = torch.randn(n, device=dev); g = torch.randn(n, device=dev)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    pr = torch.nn.Parameter(p.clone()); pr.grad = g.clone() * 0.5
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    gs = torch.tensor([0.5], device=dev)
    pb = torch.empty(n + 5, dtype=torch.bfloat16, device=dev)[:n]
    for step in (1, 2, 3):
        opt.step()
        K.adamw_step(p, g, m, v, None, 1e-3, 0.9, 0.98, 1e-6, 0.1, 1 - 0.9 ** step, 1 - 0.98 ** step, gs)
    report("adamw 3 steps", p, pr.data, 1e-5)
    out = torch.zeros(1, device=dev)
    K.sumsq(g, out)
    report("sumsq", out, (g * g).sum().reshape(1), 1e-4)


def perf():
    print("---- perf (bf16, random data) ----")
    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    for (M, N, K_) in ((48000, 1280, 1280), (48000, 3840, 1280), (48000, 5120, 1280), (48000, 1280, 5120), (4096, 1280, 1280), (4096, 51968, 1280)):
        a = bf(torch.randn(M, K_, device=dev)); b = bf(torch.randn(N, K_, device=dev))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: K.gemm_nt(a, b, out=out))
        print(f"gemm_nt {M}x{N}x{K_}: {t*1e3:.3f} ms  {2*M*N*K_/t/1e12:.1f} TF/s")
        t = timeit(lambda: torch.matmul(a, b.t(), out=out))
        print(f"   torch(hipblaslt) same: {t*1e3:.3f} ms  {2*M*N*K_/t/1e12:.1f} TF/s")
    for (R, P, Q) in ((48000, 1280, 1280), (48000, 5120, 1280), (48000, 1280, 5120)):
        a = bf(torch.randn(R, P, device=dev)); b = bf(torch.randn(R, Q, device=dev))
        out = torch.empty(P, Q, device=dev)
        t = timeit(lambda: K.gemm_tn(a, b, out=out))
        print(f"gemm_tn {R}:{P}x{Q}: {t*1e3:.3f} ms  {2*R*P*Q/t/1e12:.1f} TF/s")
    B, H, T = 32, 20, 1500
    qkv = bf(torch.randn(B, T, 3 * H * 64, device=dev))
    q, k, v = qkv[..., :1280], qkv[..., 1280:2560], qkv[..., 2560:]
    t = timeit(lambda: K.attn_fwd(q, k, v, H, False, 0.125), 10)
    fl = 4 * B * H * T * T * 64
    print(f"attn_fwd B32 H20 T1500: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s")
    o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    do = bf(torch.randn(B, T, 1280, device=dev))
    t = timeit(lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125), 10)
    print(f"attn_bwd B32 H20 T1500: {t*1e3:.3f} ms  {2.5*fl/t/1e12:.1f} TF/s (5-product flops)")
    x = bf(torch.randn(48000, 1280, device=dev)); g = torch.ones(1280, device=dev); b_ = torch.zeros(1280, device=dev)
    t = timeit(lambda: K.layernorm_fwd(x, g, b_))
    print(f"ln_fwd 48000x1280: {t*1e6:.1f} us  {2*x.numel()*2/t/1e9:.0f} GB/s")
    audio = torch.randn(32, 480000, device=dev) * 0.1
    filt = torch.rand(128, 201, device=dev)
    t = timeit(lambda: K.logmel(audio, filt), 5)
    print(f"logmel B32: {t*1e3:.3f} ms")


if __name__ == "__main__":
    print(L.load().wft_version().decode(), torch.cuda.get_device_name(0))
    for fn in (t_cast, t_ln, t_gemm_nt, t_gemm_tn, t_attn, t_embed_ce, t_audio, t_adamw):
        run(fn)
        torch.cuda.synchronize()
    nfail = sum(1 for _, ok in results if not ok)
    print(f"==== {len(results) - nfail} passed, {nfail} failed ====")
    if "--perf" in sys.argv:
        run(perf)
