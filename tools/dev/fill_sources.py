"""Developer script (GPU box): who launches the fill kernels of one training step?  Python-level: torch.zeros / zeros_like / new_zeros /
Tensor.zero_ / fill_ are wrapped and counted by calling line for ONE train_step of the chosen configuration (large-v3 at a small batch).
    python tools/dev/fill_sources.py [lora]"""
import collections
import sys
import traceback
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import bench  # noqa: E402


class A:
    model = "large-v3"


lora = len(sys.argv) > 1 and sys.argv[1] == "lora"
dev = torch.device("cuda:0")
case = bench.Case(A, dev, 0, 0, 1, False, lora=lora, muon=lora, sd=0.1 if lora else 0.0, dsa=lora)
case.measure(4, 128, 1, 2, roofline=False)
cnt = collections.Counter()


def wrap(owner, name):
    real = getattr(owner, name)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if "whisper" in x.filename or "bench.py" in x.filename]
        key = f"{Path(fr[-1].filename).name}:{fr[-1].lineno}" if fr else "torch-internal"
        cnt[(name, key)] += 1
        return real(*a, **k)

    setattr(owner, name, f)


for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n)
for n in ("zero_", "fill_", "new_zeros"):
    wrap(torch.Tensor, n)
case.measure(4, 128, 1, 0, roofline=False)
for (name, key), n in cnt.most_common(30):
    print(f"{n:6d}  {name:12s} {key}")
