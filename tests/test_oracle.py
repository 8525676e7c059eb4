"""Pins the CPU oracle (oracle/whisper_oracle.py) against the golden fixtures:
HF transformers (independent architecture implementation), HF feature extractor (log-mel),
and outputs of the REFERENCE'S OWN Python for the host-side arithmetic."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O
from tests.golden.gen_golden import ARCH_DIMS, arch_inputs, arch_params


def test_model_forward_loss_and_grads_match_hf(golden_arch):
    dims = ARCH_DIMS
    params = {k: v.requires_grad_(k != "encoder.positional_embedding") for k, v in arch_params(dims, 3).items()}
    mel, y_in, y_out = arch_inputs(dims, 11)
    logits = O.Oracle(dims, params).forward(mel, y_in)
    loss = O.cross_entropy(logits, y_out, 0.1)
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), golden_arch["logits"], rtol=2e-4, atol=2e-4)
    assert abs(loss.item() - float(golden_arch["loss"])) < 1e-5 * abs(float(golden_arch["loss"]))
    names = [str(n) for n in golden_arch["grad_norm_names"]]
    for n, ref in zip(names, golden_arch["grad_norms"]):
        got = params[n].grad.norm().item()
        assert abs(got - ref) <= 2e-4 * ref + 1e-7, (n, got, ref)
    for key in golden_arch.files:
        if key.startswith("grad::"):
            np.testing.assert_allclose(params[key[6:]].grad.numpy(), golden_arch[key], rtol=1e-3, atol=1e-6)
    # the tied embedding gets both the gather and the projection gradient
    assert "decoder.token_embedding.weight" in names


def test_cross_entropy_closed_form_equals_torch():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(3, 7, 50, generator=g) * 3
    y = torch.randint(0, 50, (3, 7), generator=g)
    y[0, :2] = -100
    for eps in (0.0, 0.05, 0.1):
        a = O.cross_entropy(logits, y, eps)
        b = O.cross_entropy_manual(logits, y, eps)
        assert abs(a.item() - b.item()) < 1e-6


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_hf_feature_extractor(golden_logmel, n_mels):
    np.testing.assert_allclose(O.mel_filters(n_mels).numpy(), golden_logmel[f"filters{n_mels}"], atol=2e-7)
    clips = []
    for i in range(2):
        a = torch.randn(O.N_SAMPLES, generator=torch.Generator().manual_seed(1234 + i)) * 0.1
        if i == 1:
            a[200000:] = 0.0
            a[:200000] *= torch.linspace(0.0, 1.0, 200000)
        clips.append(a)
    t = np.arange(O.N_SAMPLES) / 16000.0
    clips.append(torch.from_numpy((0.3 * np.sin(2 * np.pi * (200.0 + 120.0 * t) * t)).astype(np.float32)))
    mel = O.log_mel_spectrogram(torch.stack(clips), n_mels)
    assert mel.shape == (3, n_mels, 3000)
    # fp32 torch.stft vs the extractor's float64 numpy STFT: agree to 1e-4 except bins sitting on the
    # (max - 8) floor of near-silent frames
    diff = np.abs(mel[:, :, ::7].numpy() - golden_logmel[f"mel{n_mels}_sub"])
    assert np.quantile(diff, 0.999) < 2e-4 and diff.max() < 5e-3, (np.quantile(diff, 0.999), diff.max())


def test_time_warp_matches_reference(golden_host):
    spec = torch.from_numpy(golden_host["tw_spec"])
    for i in range(3):
        wp, wd = (int(v) for v in golden_host[f"tw_params{i}"])
        got = O.time_warp(spec, wp, wd)
        np.testing.assert_allclose(got.numpy(), golden_host[f"tw_out{i}"], atol=2e-6)


def test_extremes_pad_trim_match_reference(golden_host):
    lo, hi = O.extremes_lengths(float(golden_host["ext_r"]), 6, 4)
    got = O.spec_augment(torch.ones(16, 10), None, (0, 0), (0, 0), (lo, hi))
    np.testing.assert_array_equal(got.numpy(), golden_host["ext_out"])
    x = torch.from_numpy(golden_host["pad_in"])
    np.testing.assert_array_equal(O.pad_or_trim_min(x, 7).numpy(), golden_host["pad_out"])
    np.testing.assert_array_equal(O.pad_or_trim_min(x, 2).numpy(), golden_host["trim_out"])


def test_stochastic_depth_matches_reference(golden_host):
    x = torch.tensor([[1.0, -2.0, 3.0]])
    for d, ref in zip(golden_host["sd_draws"], golden_host["sd_outs"]):
        got = O.Oracle.stochastic_depth(x, lambda t: t * 2 + 1, 0.4, training=True, skip=bool(d < 0.4))
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-6)
    got = O.Oracle.stochastic_depth(x, lambda t: t * 2 + 1, 0.4, training=False, skip=False)
    np.testing.assert_allclose(got.numpy(), golden_host["sd_eval"], rtol=1e-6)


def test_deep_spec_augment_draw_order_matches_reference(golden_host):
    """blocks 0..n-2 draw (time value, time min, freq value, freq min) each; the last block is skipped."""
    torch.manual_seed(42)
    for i in range(2):
        t0, t1 = O.draw_mask_span(30, 150)
        c0, c1 = O.draw_mask_span(20, 128)
        rows = np.zeros(150, bool); rows[t0:t1] = True
        cols = np.zeros(128, bool); cols[c0:c1] = True
        np.testing.assert_array_equal(rows, golden_host[f"dsa_rows{i}"])
        np.testing.assert_array_equal(cols, golden_host[f"dsa_cols{i}"])
    assert not golden_host["dsa_rows2"].any() and not golden_host["dsa_cols2"].any()


def test_lora_invariants_of_the_reference_tests():
    """minLoRA algebra the reference pins in tests/test_lora.py:42-44,124-127,262-273,612-614,762-766."""
    g = torch.Generator().manual_seed(1)
    W = torch.randn(12, 20, generator=g)
    A, B = O.lora_init(12, 20, rank=4, generator=g)
    assert A.shape == (4, 20) and B.shape == (12, 4)
    assert B.norm() == 0 and A.norm() > 0
    assert torch.equal(O.lora_effective_weight(W, A, B, 32 / 16), W)  # B = 0 => unchanged
    B = torch.randn(12, 4, generator=g)
    s = 32 / 16
    x = torch.randn(5, 20, generator=g)
    merged = O.lora_effective_weight(W, A, B, s)
    np.testing.assert_allclose((x @ merged.T).numpy(), (x @ W.T + s * (x @ A.T) @ B.T).numpy(), atol=1e-5)
    m = (torch.rand(1, 20, generator=g) > 0.3).float() / 0.7
    np.testing.assert_allclose((x @ O.lora_effective_weight(W, A, B, s, m).T).numpy(),
                               (x @ W.T + s * ((x * m) @ A.T) @ B.T).numpy(), atol=1e-5)


def test_collate_padding_values():
    x, yi, yo = O.collate([torch.ones(2, 5), torch.ones(2, 5)], [torch.tensor([1, 2, 3]), torch.tensor([4])],
                          [torch.tensor([2, 3, 9]), torch.tensor([9])])
    assert yi.tolist() == [[1, 2, 3], [4, 0, 0]] and yo.tolist() == [[2, 3, 9], [9, -100, -100]]
