"""Developer script (GPU box): board power and shader clock (rocm-smi) while one kernel family runs back to back for a few seconds."""
import subprocess, sys, threading, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K
dev = torch.device("cuda:0")
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=20).stdout
        return out.strip().replace("\n", " | ")[:900]
    except Exception as e:
        return f"rocm-smi failed: {e}"
print("idle:", smi(), flush=True)
M = 102000
def gemm(N, Kd):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16(); o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    return lambda: K.gemm_nt(a, b, out=o)
def attn():
    B, H, T = 32, 20, 1500
    qkv = torch.randn(B, T, 3 * H * 64, device=dev).bfloat16(); d = H * 64
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    do = torch.randn(B, T, d, device=dev).bfloat16(); o, lse = K.attn_fwd(q, k, v, H, False, 0.125)
    return lambda: K.attn_bwd(q, k, v, o, lse, do, H, False, 0.125)
def ln():
    x = torch.randn(M, 1280, device=dev).bfloat16(); g = torch.ones(1280, device=dev); b = torch.zeros(1280, device=dev)
    return lambda: K.layernorm_fwd(x, g, b)
def lib(N, Kd):  # the vendor library's kernel for the same product (yard-stick)
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16(); o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bt = b.t()
    return lambda: torch.mm(a, bt, out=o)
def zeros(N, Kd):  # all-zero operands: the same instruction stream at minimum switching power
    a = torch.zeros(M, Kd, device=dev).bfloat16(); b = torch.zeros(N, Kd, device=dev).bfloat16(); o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    return lambda: K.gemm_nt(a, b, out=o)
CASES = (("gemm_nt 5120x1280", lambda: gemm(5120, 1280)), ("library 5120x1280", lambda: lib(5120, 1280)), ("gemm_nt 1280x1280", lambda: gemm(1280, 1280)),
         ("library 1280x1280", lambda: lib(1280, 1280)), ("gemm_nt 5120x1280 zeros", lambda: zeros(5120, 1280)),
         ("gemm_nt 1280x5120", lambda: gemm(1280, 5120)), ("attn_bwd enc", attn), ("layernorm_fwd", ln))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if any(k in c[0] for k in sys.argv[1:]))
for name, mk in CASES:
    f = mk()
    for _ in range(5): f()
    torch.cuda.synchronize()
    stop = [False]; samples = []
    def poll():
        while not stop[0]:
            samples.append(smi()); time.sleep(0.5)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 6.0:
        for _ in range(50): f()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop[0] = True; th.join()
    print(f"== {name}: {dt / n * 1e6:.1f} us per launch", flush=True)
    for s in samples[2:8]: print("   ", s, flush=True)
