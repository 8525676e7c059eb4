"""Developer script (GPU box): which Python lines issue the small torch ops (fills, copies, element-wise) of one optimizer step.
Runs bench.Case under a TorchDispatchMode that records the innermost package frame of every aten call.
    python tools/dev/find_launches.py [--lora] [--model large-v3] [--batch 4]"""
import sys, collections, traceback, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import torch
import bench
from torch.utils._python_dispatch import TorchDispatchMode

ap = argparse.ArgumentParser(); ap.add_argument("--lora", action="store_true"); ap.add_argument("--model", default="large-v3"); ap.add_argument("--batch", type=int, default=4)
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
for _ in range(2):
    train_step(case.net, it, case.opt, case.sched, case.t_cfg)
torch.cuda.synchronize()

sites = collections.defaultdict(collections.Counter)
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("fill", "zero", "copy", "mul", "add", "empty", "clone", "contiguous", "sum", "div", "cat", "to_copy", "rand", "uniform", "bernoulli", "normal", "dropout", "lt", "gt", "ge", "le", "where")):
            fr = [f for f in traceback.extract_stack()[:-1] if ("whisper_finetune" in f.filename or "bench.py" in f.filename)]
            where = f"{Path(fr[-1].filename).name}:{fr[-1].lineno} {fr[-1].line[:90]}" if fr else "<autograd engine / torch internals>"
            sites[name][where] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    train_step(case.net, it, case.opt, case.sched, case.t_cfg)
torch.cuda.synchronize()
for name, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
    if "empty" in name:
        continue
    print(f"== {name}: {sum(c.values())}")
    for s, n in c.most_common(10):
        print(f"   {n:5d}  {s}")
