"""Developer script (GPU box): which Python call launches a non-libwft GEMM (a Tensile 'Cijk_*' kernel) in BASELINE configs[2]."""
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import bench  # noqa: E402


class A:
    model = "large-v3"
    lora = True
    muon = True
    stochastic_depth = 0.1
    deep_spec_augment = True


dev = torch.device("cuda:0")
case = bench.Case(A, dev, 0, 0, 1, False, lora=True, muon=True, sd=0.1, dsa=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
case.measure(B, 128, 1, 2, roofline=False)
import collections
import traceback

from torch.overrides import TorchFunctionMode

sites = collections.Counter()


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name in ("matmul", "__matmul__", "mm", "linear", "bmm", "addmm", "__rmatmul__", "einsum", "baddbmm"):
            fr = [f"{f.filename.split('/')[-1]}:{f.lineno}:{f.name}" for f in traceback.extract_stack(limit=7)[:-1]]
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
            sites[(name, " < ".join(reversed(fr[-4:])), shapes)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    case.measure(B, 128, 1, 0, roofline=False)
torch.cuda.synchronize()
for (name, where, shapes), n in sites.most_common(12):
    print(n, name, shapes, "\n      ", where)
