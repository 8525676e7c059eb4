"""Rows T and f4 pinned to the reference itself: fixtures produced by the REFERENCE'S OWN
`train_step` and `AudioDataset` (tests/golden/gen_golden.py: gen_ref_train_step, gen_ref_dataset), replayed through this
package's host code on the CPU.  The GPU leg of the train_step golden is tests/test_model_gpu.py."""
import json
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.golden.gen_golden import (ARCH_DIMS, TRAIN_STEP_CFG, TRAIN_STEP_OPT, GoldenRecords, GoldenTokenizer, OracleModule,
                                     arch_params, dataset_cases, train_step_case)
from whisper_finetune.data.data_loader import N_FRAMES, AudioDataset
from whisper_finetune.model.model_utils import train_step
from whisper_finetune.model.scheduler import get_scheduler

GOLD = Path(__file__).parent / "golden"


def test_train_step_reproduces_the_references_loss_sequence():
    """Same model (the CPU oracle as an nn.Module), same batches, same optimizer: this package's train_step must give the
    reference's four losses, learning rates and final parameters — the accumulation / scaling / clip / step / schedule order
    of model/model_utils.py:54-125 is what is being compared (fp32 on the CPU: exact up to summation order)."""
    ref = np.load(GOLD / "ref_train_step.npz")
    model = OracleModule(ARCH_DIMS, arch_params(ARCH_DIMS, seed=3))
    opt = torch.optim.AdamW(model.parameters(), **TRAIN_STEP_OPT)
    sched = get_scheduler(opt, {"type": "linear", "warmup_steps": 2}, 4)
    it = iter(train_step_case())
    losses, lrs = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(1, 5):
            lrs.append(opt.param_groups[0]["lr"])
            losses.append(train_step(model, it, opt, sched, dict(TRAIN_STEP_CFG), step=step))
    np.testing.assert_allclose(lrs, ref["lrs"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(losses, ref["losses"], rtol=1e-6)
    for k, v in model.state().items():
        np.testing.assert_allclose(v.detach().double().norm().item(), ref["final_norm/" + k], rtol=1e-6, err_msg=k)


_DS = json.loads((GOLD / "ref_dataset.json").read_text())


@pytest.mark.parametrize("case", _DS["runs"], ids=lambda c: f"seed{c['seed']}")
def test_audio_dataset_item_matches_the_reference(case):
    """Token / target construction, the partial-segment cut, the augmentation draws (their values where the reference's
    output reveals them, and their number/order through the next default-generator value)."""
    records, _ = dataset_cases()
    ds = AudioDataset(GoldenRecords(records), GoldenTokenizer(), n_mels=80, **case["kw"])
    torch.manual_seed(case["seed"])
    audio, y_in, y_out, params, ext, cut = ds[case["index"]]
    nxt = torch.rand(1).item()
    assert y_in.tolist() == case["y_in"]
    assert y_out.tolist() == case["y_out"]
    assert nxt == case["next_rand"]
    assert sorted(ds.invalid_indices) == case["invalid"]
    assert audio.shape == (480000,) and audio.dtype == torch.float32
    if case["kept_frames"] is not None:
        assert cut == case["kept_frames"]
    apply, _, _, t0, t1, f0, f1, _ = params.tolist()
    lo, hi = ext.tolist()
    if case["kw"].get("spec_augment") and apply:
        # the reference's output shows the masked spans (a warped ramp has no zero of its own away from the edges)
        if case["zero_time"] is not None and t1 > t0:
            assert case["zero_time"][0] <= t0 and t1 <= case["zero_time"][1]
            assert t1 - t0 >= (case["zero_time"][1] - case["zero_time"][0]) - 2
        expect = set(range(f0, f1)) | set(range(0, lo)) | set(range(80 - hi, 80))
        assert sorted(expect) == case["zero_mels"]
    elif case["kw"].get("spec_augment"):
        assert case["zero_mels"] == sorted(set(range(0, lo)) | set(range(80 - hi, 80)))
    else:
        assert apply == 0 and case["zero_mels"] == [] and case["zero_time"] is None


def test_invalid_timestamp_raises_like_the_reference():
    (kind, msg), = _DS["errors"]
    ds = AudioDataset(GoldenRecords([{"audio": {"array": torch.zeros(16000)}, "text": "<|0.01|>odd", "language": "de", "prompt": ""}]),
                      GoldenTokenizer(), n_mels=80, no_timestamps_rate=0.0, prompt_use_rate=0.0)
    with pytest.raises(ValueError) as err:
        ds[0]
    assert kind == "ValueError" and str(err.value) == msg


def test_scheduler_public_constructors_match_get_scheduler():
    from whisper_finetune.model.scheduler import get_cosine_annealing_with_warmup_restarts

    gold = json.loads((GOLD / "ref_sched.json").read_text())["cosine_with_warmup_restarts"]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    s = get_cosine_annealing_with_warmup_restarts(opt, gold["conf"]["warmup_steps"], 120, num_cycles=gold["conf"]["lr_num_cycles"],
                                                  gamma=gold["conf"]["lr_gamma"])
    for ref in gold["lrs"]:
        assert abs(opt.param_groups[0]["lr"] - ref) < 1e-12
        opt.step(); s.step()
