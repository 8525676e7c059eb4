"""Developer script (GPU box): attn_fwd_pipe_kernel (variant 2) against attn_fwd_kernel (variant 1): bit identity of o / lse and
time per call on the model's shapes."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
dev = torch.device("cuda:0"); lib = L.load()
torch.manual_seed(0)
def run(v, q, k, vv, H, causal):
    old = K.set_variant("fwd", v)
    try:
        for _ in range(3): o, lse = K.attn_fwd(q, k, vv, H, causal, 0.125)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(4):
            e0.record()
            for _ in range(5): o, lse = K.attn_fwd(q, k, vv, H, causal, 0.125)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        return o, lse, min(ts)
    finally:
        K.set_variant("fwd", old)
for B, H, Tq, Tk, causal in ((32, 20, 1500, 1500, False), (68, 20, 1500, 1500, False), (87, 20, 1500, 1500, False), (68, 20, 128, 128, True),
                             (68, 20, 128, 1500, False), (32, 20, 448, 448, True), (32, 20, 448, 1500, False), (3, 6, 50, 1500, False),
                             (2, 8, 77, 77, True), (2, 6, 1, 1, True), (4, 8, 1500, 1500, False), (1, 20, 200, 190, False)):
    d = H * 64
    q = torch.randn(B, Tq, d, device=dev).bfloat16() * 2
    kv = torch.randn(B, Tk, 2 * d, device=dev).bfloat16() * 2
    k, v = kv[..., :d], kv[..., d:]
    o1, l1, t1 = run(1, q, k, v, H, causal)
    o2, l2, t2 = run(2, q, k, v, H, causal)
    print(f"B={B} H={H} Tq={Tq} Tk={Tk} causal={causal}: 8-wave {t1:.1f} us | pipelined {t2:.1f} us ({t1 / t2:.3f}x) | o equal {torch.equal(o1, o2)} "
          f"lse equal {torch.equal(l1, l2)} max|do| {(o1.float() - o2.float()).abs().max().item():.2e}", flush=True)
