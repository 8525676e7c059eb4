"""Developer script (GPU box): the one-wave-per-SIMD NT kernel (gemm_nt4w) against the ping-pong kernel and hipBLASLt.
  python tools/dev/nt4w_ab.py check     # correctness vs an fp32 torch reference, all epilogues, ragged M
  python tools/dev/nt4w_ab.py time      # interleaved timing on the encoder shapes (one process, random data)
hipBLASLt (torch.matmul) is the yard-stick only; the product path never calls it."""
import sys, time, ctypes as C
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0")
lib = L.load()
def variant(v): K.set_variant("nt", v)  # 0 = 4w where eligible, 1 = ping-pong
bf = lambda x: x.to(torch.bfloat16)

def gelu(x): return torch.nn.functional.gelu(x)
def dgelu(x):
    x = x.double()
    return (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5).float()

def check():
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, Kd) in ((1024, 256, 768), (2048 + 112, 512, 896), (4000, 1280, 1280), (3072, 256, 5120), (1500 * 3, 3840, 1280)):
        a = bf(torch.randn(M, Kd, device=dev)); b = bf(torch.randn(N, Kd, device=dev) * 0.05)
        bias = torch.randn(N, device=dev); res = bf(torch.randn(M, N, device=dev)); auxin = bf(torch.randn(M, N, device=dev))
        ref0 = a.float() @ b.float().t()
        cases = [("none", dict(), ref0),
                 ("bias", dict(bias=bias), ref0 + bias),
                 ("bias+res", dict(bias=bias, residual=res), ref0 + bias + res.float()),
                 ("alpha.beta", dict(alpha=0.5, residual=res, beta=2.0), 0.5 * ref0 + 2.0 * res.float()),
                 ("gelu", dict(bias=bias, epilogue=L.EPI_GELU), gelu(ref0 + bias)),
                 ("dgelu", dict(epilogue=L.EPI_DGELU, aux=auxin), ref0 * dgelu(auxin.float())),
                 ("mul_aux", dict(epilogue=L.EPI_MUL_AUX, aux=auxin), ref0 * auxin.float()),
                 ("period", dict(bias=bias, valid_rows_period=1504 if M > 2000 else 128, valid_rows=1500 if M > 2000 else 100), None)]
        for v in (0, 1):
            variant(v)
            for name, kw, ref in cases:
                out = K.gemm_nt(a, b, **kw)
                if name == "period":
                    per, val = kw["valid_rows_period"], kw["valid_rows"]
                    ref = (ref0 + bias).clone(); rows = torch.arange(M, device=dev) % per >= val; ref[rows] = 0
                err = ((out.float() - ref).norm() / ref.norm()).item()
                worst = max(worst, err)
                flag = "" if err < 4e-3 else "  <-- FAIL"
                print(f"v{v} {M}x{N}x{Kd} {name:10s} rel {err:.2e}{flag}", flush=True)
            # GELU_GRAD: two outputs
            aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            out = K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)
            e1 = ((out.float() - gelu(ref0 + bias)).norm() / gelu(ref0 + bias).norm()).item()
            e2 = ((aux.float() - dgelu(ref0 + bias)).norm() / dgelu(ref0 + bias).norm()).item()
            print(f"v{v} {M}x{N}x{Kd} gelu_grad  rel {e1:.2e} {e2:.2e}" + ("" if max(e1, e2) < 4e-3 else "  <-- FAIL"), flush=True)
            worst = max(worst, e1, e2)
            # GELU with the pre-activation written to aux
            aux2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            out = K.gemm_nt(a, b, bias=bias, epilogue=L.EPI_GELU, aux=aux2)
            e3 = ((aux2.float() - (ref0 + bias)).norm() / (ref0 + bias).norm()).item()
            print(f"v{v} {M}x{N}x{Kd} gelu+pre   rel {e3:.2e}" + ("" if e3 < 4e-3 else "  <-- FAIL"), flush=True)
            worst = max(worst, e3)
            # fused column sums
            cs = torch.empty(N, device=dev)
            out = K.gemm_nt(a, b, epilogue=L.EPI_MUL_AUX, aux=auxin, colsum=cs)
            refcs = (ref0 * auxin.float()).sum(0)  # (the fused sums are taken before the bf16 rounding of C)
            e4 = ((cs - refcs).norm() / refcs.norm()).item()
            print(f"v{v} {M}x{N}x{Kd} colsum     rel {e4:.2e}" + ("" if e4 < 1e-4 else "  <-- FAIL"), flush=True)
            worst = max(worst, e4 * 10)
        # the two kernels against each other + reproducibility of the new one
        variant(0); o0 = K.gemm_nt(a, b, bias=bias); o0b = K.gemm_nt(a, b, bias=bias)
        variant(1); o1 = K.gemm_nt(a, b, bias=bias)
        print(f"   4w == 4w again: {torch.equal(o0, o0b)}   4w vs pp: max abs diff {(o0.float() - o1.float()).abs().max().item():.3e}", flush=True)
    # batch > 1
    a = bf(torch.randn(3 * 1024, 256, device=dev)); b = bf(torch.randn(2 * 256, 256, device=dev))
    variant(0)
    out = K.gemm_nt(a, b, M=1024, N=256, K=256, batch=3, strideA=1024 * 256, strideB=0)
    ref = torch.cat([a[i * 1024:(i + 1) * 1024].float() @ b[:256].float().t() for i in range(3)])
    print("batch rel", ((out.float() - ref).norm() / ref.norm()).item())
    print("WORST", worst, "OK" if worst < 4e-3 else "FAIL")

def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def timing(shapes=None, rounds=5):
    torch.manual_seed(0)
    print("| shape (M x N x K) | epilogue | pp TF/s | 4w TF/s | hipBLASLt TF/s | 4w/pp | 4w/lib |")
    print("|---|---|---|---|---|---|---|")
    for M in (102000, 48000):
        for N, Kd in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
            a = bf(torch.randn(M, Kd, device=dev)); b = bf(torch.randn(N, Kd, device=dev) * 0.03)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            bias = torch.randn(N, device=dev)
            res = bf(torch.randn(M, N, device=dev)); aux = bf(torch.randn(M, N, device=dev))
            for ename, kw in (("bias", dict(bias=bias)), ("bias+res", dict(bias=bias, residual=res)),
                              ("gelu_grad", dict(bias=bias, epilogue=L.EPI_GELU_GRAD, aux=aux)), ("mul_aux", dict(epilogue=L.EPI_MUL_AUX, aux=aux))):
                if ename == "gelu_grad" and N != 5120: continue
                if ename == "mul_aux" and N != 5120: continue
                if ename == "bias+res" and N != 1280: continue
                f = lambda: K.gemm_nt(a, b, out=out, **kw)
                fl = 2.0 * M * N * Kd
                tp, t4, tl = [], [], []
                for v in (0, 1): variant(v); f(); f()
                torch.matmul(a, b.t(), out=out)
                for _ in range(rounds):
                    variant(1); tp.append(timeit(f))
                    variant(0); t4.append(timeit(f))
                    if ename == "bias": tl.append(timeit(lambda: torch.matmul(a, b.t(), out=out)))
                med = lambda x: sorted(x)[len(x) // 2]
                lp, l4 = fl / med(tp) / 1e12, fl / med(t4) / 1e12
                ll = fl / med(tl) / 1e12 if tl else float("nan")
                print(f"| {M} x {N} x {Kd} | {ename} | {lp:.0f} | {l4:.0f} | {ll:.0f} | {l4/lp:.3f} | {l4/ll:.3f} |", flush=True)

if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check": check()
    else: timing()
