"""Developer script (GPU box): who launches the ~830 FillFunctor kernels per headline step (2.7 ms in profiles/r04_d)?  One train_step
of large-v3 at a small batch under torch.profiler with Python stacks; aten::zero_ / aten::fill_ / aten::zeros calls grouped by the
innermost frame inside this repository."""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import bench  # noqa: E402


class A:
    model = "large-v3"


dev = torch.device("cuda:0")
case = bench.Case(A, dev, 0, 0, 1, False)
case.measure(4, 128, 1, 2, roofline=False)  # warm-up: shadows, homes, tables


def one():
    case.measure(4, 128, 1, 0, roofline=False)


from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    one()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::full"):
        frame = next((s for s in ev.stack if "/whisper" in s or "bench.py" in s or "torch/nn/parallel" in s or "optim" in s), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, frame)] += 1
for (name, frame), n in cnt.most_common(30):
    print(f"{n:6d}  {name:18s} {frame}")
