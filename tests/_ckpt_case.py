"""The two tiny checkpoints of the §8(f3) round-trip tests, written by THIS package's save_model (shared by
tests/golden/gen_golden.py, which pushes them through the reference's converter / merge flow, and by the tests, which rebuild
them and compare with the committed results)."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
LORA_CFG = {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.1}


def build(tmpdir: Path, dims, params):
    """-> (full.pt, lora.pt): a full fine-tune checkpoint and a LoRA checkpoint (parametrized keys, non-zero B) of the same base."""
    if str(ROOT / "whisper-finetune_amd") not in sys.path:
        sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.model import lora as lora_mod
    from whisper_finetune.model.model_utils import save_model

    edims = ModelDimensions(**{k: getattr(dims, k) for k in ModelDimensions.__dataclass_fields__})
    model = Whisper(edims)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    full = tmpdir / "full.pt"
    save_model(model, str(full))
    torch.manual_seed(17)  # lora_A: kaiming_uniform_ from the global generator
    lora_mod.apply_lora(model, dict(LORA_CFG))
    g = torch.Generator().manual_seed(23)
    for mod in model.modules():
        if "parametrizations" in mod._modules:
            ad = mod.parametrizations.weight[0]
            with torch.no_grad():
                ad.lora_B.copy_(torch.randn(ad.lora_B.shape, generator=g) * 0.05)
    lora = tmpdir / "lora.pt"
    save_model(model, str(lora))
    return full, lora
