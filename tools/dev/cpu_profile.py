"""Developer script (GPU box): host-side cost of one optimizer step (cProfile, cumulative), for launch-bound configurations.
    python tools/dev/cpu_profile.py [--model base] [--batch 8] [--lora]"""
import sys, argparse, cProfile, pstats, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--model", default="base"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--lora", action="store_true")
a = ap.parse_args()
class A: pass
args = A(); args.model = a.model
dev = torch.device("cuda:0")
case = bench.Case(args, dev, 0, 0, 1, False, lora=a.lora, muon=a.lora, sd=0.1 if a.lora else 0.0, dsa=a.lora)
from whisper_finetune.model.model_utils import train_step
B, S = a.batch, 128
audio = torch.randn(B, 480000, device=dev) * 0.1
y_in, y_out = bench.synthetic_tokens(B, S, dev, 0)
def batches():
    while True:
        yield case.frontend(audio, training=True), y_in, y_out
it = batches()
for _ in range(5):
    train_step(case.net, it, case.opt, case.sched, case.t_cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    train_step(case.net, it, case.opt, case.sched, case.t_cfg)
torch.cuda.synchronize(); print(f"wall {1e3 * (time.perf_counter() - t0) / 10:.2f} ms/step")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):  # the backward pass on this thread: visible to cProfile
    pr.enable()
    for _ in range(10):
        train_step(case.net, it, case.opt, case.sched, case.t_cfg)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
