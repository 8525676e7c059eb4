"""Developer script (GPU box): whole-model parity of the libwft engine against the CPU oracle
(tiny dims), then a first large-v3 step timing.   python tools/dev/model.py [--perf]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
sys.path.insert(0, str(ROOT))
from oracle import whisper_oracle as O  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine.whisper_model import MODEL_DIMS, Whisper, init_random_  # noqa: E402
from whisper_finetune.model import lora as lora_mod  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def parity(name="tiny", B=2, S=24, eps=0.1, with_lora=False):
    dims = O.DIMS[name]
    params = O.init_params(dims, seed=0)
    # larger weights than 0.02 std make the comparison more demanding
    g = torch.Generator().manual_seed(5)
    for k, v in params.items():
        if k.endswith("bias"):
            params[k] = torch.randn(v.shape, generator=g) * 0.02
        elif "ln" in k and k.endswith("weight"):
            params[k] = 1 + torch.randn(v.shape, generator=g) * 0.05
    audio, y_in, y_out = O.synthetic_batch(dims, B, S)
    y_out[0, :3] = -100
    mel = O.log_mel_spectrogram(audio, dims.n_mels)
    model = Whisper(MODEL_DIMS[name])
    model.load_state_dict(params)
    lora_cfg = None
    if with_lora:
        lora_mod.apply_lora(model, {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.0})
        gl = torch.Generator().manual_seed(9)
        lora_cfg = {}
        for n, m in model.named_modules():
            if hasattr(m, "parametrizations"):
                ad = m.parametrizations.weight[0]
                with torch.no_grad():
                    ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=gl) * 0.05)
                lora_cfg[n] = (ad.lora_A.detach().clone().requires_grad_(True), ad.lora_B.detach().clone().requires_grad_(True), ad.scaling, None)
    model.to(dev).train()
    # oracle with autograd
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding" and not with_lora) for k, v in params.items()}
    orc = O.Oracle(dims, p_req, lora=lora_cfg)
    logits_ref = orc.forward(mel, y_in)
    loss_ref = O.cross_entropy(logits_ref, y_out, eps)
    loss_ref.backward()
    # engine
    mel_g = K.logmel(audio.to(dev), O.mel_filters(dims.n_mels).to(dev))
    print(f"[{name} lora={with_lora}] logmel rel err {rel(mel_g, mel):.3e}  max abs {(mel_g.cpu()-mel).abs().max():.3e}")
    loss = model(mel.to(dev), y_in.to(dev), targets=y_out.to(dev), label_smoothing=eps)
    loss.backward()
    print(f"  loss engine {loss.item():.6f}  oracle {loss_ref.item():.6f}  rel {abs(loss.item()-loss_ref.item())/loss_ref.item():.3e}")
    with torch.no_grad():
        model.eval()
        logits = model(mel.to(dev), y_in.to(dev))
        model.train()
    print(f"  logits rel-L2 {rel(logits, logits_ref):.3e}  argmax agree {(logits.argmax(-1).cpu() == logits_ref.argmax(-1)).float().mean():.4f}")
    worst = []
    if with_lora:
        for n, (A, Bm, s, _) in lora_cfg.items():
            m = dict(model.named_modules())[n].parametrizations.weight[0]
            worst.append((rel(m.lora_A.grad, A.grad), n + ".lora_A"))
            worst.append((rel(m.lora_B.grad, Bm.grad), n + ".lora_B"))
        assert all(p.grad is None for n_, p in model.named_parameters() if "lora" not in n_)
    else:
        for n, p in model.named_parameters():
            worst.append((rel(p.grad, p_req[n].grad), n))
    worst.sort(reverse=True)
    print("  worst grad rel-L2:", ", ".join(f"{n}={e:.2e}" for e, n in worst[:6]))
    print(f"  median grad rel-L2: {sorted(e for e, _ in worst)[len(worst)//2]:.3e}")


def perf(name="large-v3", B=32, S=128, steps=3):
    torch.manual_seed(0)
    dims = MODEL_DIMS[name]
    with torch.device(dev):
        model = Whisper(dims)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0, 0.02)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
        from whisper_finetune.engine.whisper_model import sinusoids
        model.encoder.positional_embedding.copy_(sinusoids(dims.n_audio_ctx, dims.n_audio_state))
        model.decoder.mask.copy_(torch.empty(dims.n_text_ctx, dims.n_text_ctx).fill_(float("-inf")).triu_(1))
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True)
    mel = torch.randn(B, dims.n_mels, 3000, device=dev)
    y_in = torch.randint(0, 50000, (B, S), device=dev)
    y_out = torch.randint(0, 50000, (B, S), device=dev)
    for i in range(steps + 2):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = model(mel, y_in, targets=y_out, label_smoothing=0.1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"[{name}] B={B} S={S}: {dt*1e3:.1f} ms/step  {B*30/dt:.0f} audio-s/s  loss {loss.item():.4f}  "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")


if __name__ == "__main__":
    if "--noparity" not in sys.argv:
        parity("tiny", with_lora=False)
        parity("tiny", with_lora=True)
    if "--perf" in sys.argv:
        perf("large-v3", 8, 128)
        perf("large-v3", 32, 128)
