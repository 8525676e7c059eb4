"""the checkpoint / merge wire format, round-tripped through the REFERENCE'S OWN tools.

tests/golden/gen_golden.py (`gen_ref_checkpoint`, build container only) writes two checkpoints of a tiny model with THIS
package's `save_model`, then — in a process that imports the reference's package —
  * pushes the full fine-tune checkpoint through `scripts/convert_openai_to_hf.py:172-224` (convert_openai_whisper_to_tfms) and
    records the resulting HF model's logits / loss;
  * pushes the LoRA checkpoint (parametrized keys incl. `lora_dropout_mask`) through the flow of
    `scripts/merge_lora_weights.py:26-60` (apply_lora -> load_state_dict, raising on any missing / unexpected key ->
    merge_lora) and records the merged weights.
Here the same checkpoints are rebuilt and compared: the oracle's forward on the saved weights must give the converter's logits,
and this package's merge_lora must give the reference's merged weights (CPU: the torch path of LoRAParametrization.forward; the
`-m gpu` twin runs the wft_lora_merge kernel)."""
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _ckpt_case  # noqa: E402
from golden.gen_golden import CKPT_DIMS, arch_inputs, arch_params  # noqa: E402

FIX = np.load(HERE / "golden" / "ref_checkpoint.npz")


@pytest.fixture(scope="module")
def ckpts():
    d = Path(tempfile.mkdtemp())
    return _ckpt_case.build(d, CKPT_DIMS, arch_params(CKPT_DIMS, 3))


def _load_lora_model(path, device="cpu"):
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.model import lora as lora_mod

    ck = torch.load(path, map_location="cpu", weights_only=True)
    m = Whisper(ModelDimensions(**ck["dims"]))
    lora_mod.apply_lora(m, dict(_ckpt_case.LORA_CFG))
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in ck["model_state_dict"].items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.to(device).eval(), lora_mod


def test_saved_checkpoint_through_the_reference_converter_gives_the_oracle_logits(ckpts):
    full, _ = ckpts
    ck = torch.load(full, map_location="cpu", weights_only=True)
    assert set(ck) == {"model_state_dict", "dims"} and ck["dims"]["n_vocab"] == CKPT_DIMS.n_vocab
    sd = ck["model_state_dict"]
    assert all(v.dtype == torch.float16 for v in sd.values() if v.is_floating_point())  # save_model: fp16 (model_utils.py:130-135)
    params = {k: v.float() for k, v in sd.items()}
    mel, y_in, y_out = arch_inputs(CKPT_DIMS, 11)
    with torch.no_grad():
        logits = O.Oracle(CKPT_DIMS, params).forward(mel, y_in)
    ref = torch.from_numpy(FIX["hf_logits_s17"])
    got = logits[:, :, ::17]
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing=0.1).item()
    assert abs(loss - float(FIX["hf_loss"])) <= 1e-5 * abs(float(FIX["hf_loss"]))
    assert np.array_equal(logits.argmax(-1).numpy(), FIX["hf_argmax"])  # token ids: bit-exact


def test_lora_checkpoint_keys_and_merge_match_the_reference_flow(ckpts):
    _, lora = ckpts
    ck = torch.load(lora, map_location="cpu", weights_only=True)
    keys = sorted(k for k in ck["model_state_dict"] if "lora" in k)
    assert keys == list(FIX["lora_keys"])  # parametrizations.weight.0.lora_A / lora_B / lora_dropout_mask under the reference's names
    assert any(k.endswith("lora_dropout_mask") for k in keys)
    m, lora_mod = _load_lora_model(lora)
    lora_mod.merge_lora(m)
    assert not lora_mod.is_lora_enabled(m)
    sd = m.state_dict()
    for name in FIX["linear_names"]:
        ref = torch.from_numpy(FIX["merged::" + str(name)])
        got = sd[str(name) + ".weight"]
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 1e-6 * ref.abs().max().item() + 1e-7, name


@pytest.mark.gpu
def test_kernel_merge_matches_the_reference_flow(ckpts):
    _, lora = ckpts
    m, lora_mod = _load_lora_model(lora, "cuda:0")
    lora_mod.merge_lora(m)  # W.is_cuda: wft_lora_merge
    sd = m.state_dict()
    for name in FIX["linear_names"]:
        ref = torch.from_numpy(FIX["merged::" + str(name)])
        got = sd[str(name) + ".weight"].cpu()
        assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-7, name
