"""Developer script (GPU box): the one-wave-per-SIMD weight-gradient GEMM (gemm_tn4w) against gemm_tn256 and fp32 torch math.
  python tools/dev/tn4w_ab.py check | time"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0"); lib = L.load()
def variant(v): K.set_variant("tn", v)
bf = lambda x: x.to(torch.bfloat16)

def check():
    torch.manual_seed(0)
    worst = 0.0
    for (R, P, Q) in ((16384, 256, 256), (20000 + 37, 512, 256), (48000, 1280, 1280), (102000, 1280, 5120), (30000 + 63, 3840, 1280), (6400, 5120, 1280)):
        for rep in range(2):
            a = bf(torch.randn(R, P, device=dev)); b = bf(torch.randn(R, Q, device=dev))
            ref = (a.double().t() @ b.double()).float() if R * P * Q < 2e12 else a.float().t() @ b.float()
            outs = []
            for v in (0, 1):
                variant(v)
                o = K.gemm_tn(a, b)
                outs.append(o)
                e = ((o - ref).norm() / ref.norm()).item(); worst = max(worst, e)
                print(f"v{v} R={R} {P}x{Q} rel {e:.2e}" + ("" if e < 2e-5 else "  <-- FAIL"), flush=True)
            variant(0)
            o2 = K.gemm_tn(a, b)
            acc = o2.clone(); K.gemm_tn(a, b, out=acc, accumulate=True)
            e2 = ((acc - 2 * ref).norm() / (2 * ref).norm()).item(); worst = max(worst, e2)
            print(f"   4w again bitwise: {torch.equal(outs[0], o2)}  accumulate rel {e2:.2e}  4w-vs-pp max abs {(outs[0]-outs[1]).abs().max().item():.3e}", flush=True)
    print("WORST", worst, "OK" if worst < 2e-5 else "FAIL")

def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def timing():
    torch.manual_seed(0)
    print("| R | P x Q | pp TF/s | 4w TF/s | 4w/pp |"); print("|---|---|---|---|---|")
    for R in (102000, 48000):
        for P, Q in ((1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)):
            a = bf(torch.randn(R, P, device=dev)); b = bf(torch.randn(R, Q, device=dev)); out = torch.empty(P, Q, device=dev)
            f = lambda: K.gemm_tn(a, b, out=out)
            for v in (0, 1): variant(v); f(); f()
            tp, t4 = [], []
            for _ in range(5):
                variant(1); tp.append(timeit(f)); variant(0); t4.append(timeit(f))
            med = lambda x: sorted(x)[len(x) // 2]
            fl = 2.0 * R * P * Q
            print(f"| {R} | {P} x {Q} | {fl/med(tp)/1e12:.0f} | {fl/med(t4)/1e12:.0f} | {med(tp)/med(t4):.3f} |", flush=True)

if __name__ == "__main__":
    (check if (len(sys.argv) < 2 or sys.argv[1] == "check") else timing)()
