"""Developer script (GPU box): per-tensor gradient error of the engine against the fp32 oracle and against its bf16-emulation
mode on the small models (calibration of the tolerances in tests/test_model_gpu.py)."""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
from oracle import whisper_oracle as O
from whisper_finetune.engine import kernels as K
from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
DEV = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()
for name, B, S in (("tiny", 2, 24), ("base", 2, 32)):
    dims = O.DIMS[name]
    params = O.init_params(dims, seed=3)
    g = torch.Generator().manual_seed(5)
    for k, v in params.items():
        if k.endswith("bias"): params[k] = torch.randn(v.shape, generator=g) * 0.02
        elif "ln" in k and k.endswith("weight"): params[k] = 1 + torch.randn(v.shape, generator=g) * 0.05
    audio, y_in, y_out = O.synthetic_batch(dims, B, S)
    y_out[0, :3] = -100
    m = Whisper(ModelDimensions(**vars(dims))); m.load_state_dict(params); m.to(DEV).train()
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1); loss.backward()
    for emu in (True, False):
        p = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
        l = O.cross_entropy(O.Oracle(dims, p, emulate_bf16=emu).forward(mel.cpu(), y_in), y_out, 0.1); l.backward()
        errs = {n: rel(q.grad, p[n].grad) for n, q in m.named_parameters()}
        import re
        rest = {n: e for n, e in errs.items() if not re.search(r"decoder\.blocks\.\d+\.attn\.(query|key)\.", n)}
        print("   max outside decoder self-attention q/k:", sorted(rest.items(), key=lambda kv: -kv[1])[:2])
        w = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(name, "emulated" if emu else "fp32", "loss rel", abs(loss.item() - l.item()) / l.item(), "max", w[0], "median", float(np.median(list(errs.values()))))
