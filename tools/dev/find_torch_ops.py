"""Developer script (GPU box): every torch-level op of one steady-state step that touches a large device tensor, by call site —
what is NOT a libwft kernel in a configuration's step.    python tools/dev/find_torch_ops.py [lora|full|turbo] [batch]"""
import collections
import sys
import traceback
from pathlib import Path

import torch
from torch.overrides import TorchFunctionMode

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lora"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8


class A:
    model = "large-v3"


dev = torch.device("cuda:0")
if what == "lora":
    case = bench.Case(A, dev, 0, 0, 1, False, lora=True, muon=True, sd=0.1, dsa=True)
else:
    case = bench.Case(A, dev, 0, 0, 1, False)
case.measure(B, 128, 1, 2, roofline=False)
sites = collections.Counter()
SKIP = {"size", "dim", "stride", "data_ptr", "is_contiguous", "numel", "__get__", "view", "reshape", "detach", "expand", "t", "transpose",
        "__getitem__", "untyped_storage", "is_floating_point", "storage_offset", "element_size", "contiguous", "requires_grad_", "unbind",
        "__set__", "_backward_hooks", "apply", "backward", "record_stream", "type", "is_cuda", "__len__", "flatten", "squeeze", "unsqueeze",
        "view_as", "permute", "narrow", "split", "chunk", "__hash__", "__repr__", "is_pinned", "is_alias_of", "_is_view"}


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            big = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor) and a.is_cuda and a.numel() >= (1 << 20)]
            if big:
                fr = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack(limit=6)[:-1]]
                sites[(name, " < ".join(reversed(fr[-3:])), tuple(big[0].shape), str(big[0].dtype)[6:])] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    case.measure(B, 128, 1, 0, roofline=False)
torch.cuda.synchronize()
for (name, where, shape, dt), n in sites.most_common(40):
    print(f"{n:5d} {name:18s} {str(shape):22s} {dt:9s} {where}")
