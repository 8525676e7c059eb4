"""dev tool: which Python lines launch the small fill / copy kernels of one training step (torch profiler, with stacks)."""
import sys, collections, torch
sys.path.insert(0, "whisper-finetune_amd")
sys.path.insert(0, ".")
from whisper_finetune.engine.whisper_model import Whisper, MODEL_DIMS
from whisper_finetune.model.optimizer import WftAdamW

dev = torch.device("cuda:0")
dims = MODEL_DIMS["base"]
torch.manual_seed(0)
with torch.device(dev):
    model = Whisper(dims)
model.train()
opt = WftAdamW(model.parameters(), lr=1e-5)
B, S = 4, 32
mel = torch.randn(B, dims.n_mels, 3000, device=dev)
y_in = torch.randint(0, 50000, (B, S), device=dev)
y_out = torch.randint(0, 50000, (B, S), device=dev)

def step():
    loss = model(mel, y_in, targets=y_out, label_smoothing=0.1)
    loss.backward()
    opt.fuse_clip_grad_norm(1.0) if hasattr(opt, "fuse_clip_grad_norm") else None
    opt.step(); opt.zero_grad(set_to_none=True)

for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::zeros"):
        st = [s for s in (ev.stack or []) if "whisper_finetune" in s or "autograd" in s][:2]
        cnt[(ev.name, tuple(st))] += 1
for (name, st), n in cnt.most_common(30):
    print(n, name, " <- ".join(st))
