"""V row: evaluation host logic against values produced by the REFERENCE'S OWN eval code (tests/golden/ref_eval.json),
the WER/CER edge cases the reference pins (tests/test_metrics.py:21-77,111-146), and — on the GPU — the fused
token-statistics kernel and the evaluator end to end."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from whisper_finetune.eval import metrics as M
from whisper_finetune.eval.utils import VOCAB_SPECS, normalize_text

GOLD = json.loads((Path(__file__).parent / "golden" / "ref_eval.json").read_text())


def test_normalizer_matches_reference():
    for v, cases in GOLD["normalize"].items():
        for text, expect in cases:
            assert normalize_text(text, **VOCAB_SPECS[v]) == expect


def test_wer_cer_definitions_and_edge_cases():
    assert M.wer("hello world", "hello world") == 0.0
    assert M.wer("hello world", "hello there") == 0.5
    assert M.wer("a b c d", "a b c d e") == 0.25  # one insertion over 4 reference words
    assert abs(M.cer("abcd", "abd") - 0.25) < 1e-12
    assert M.compute_wer(["", "x", "a b"], ["", "", "a c"]) == [0.0, 1.0, 0.5]
    assert M.compute_cer_batch(["", "x"], ["", " "]) == [0.0, 1.0]


def test_ece_matches_reference():
    e = GOLD["ece"]
    assert abs(M.compute_ece(e["conf"], e["ok"]) - e["value"]) < 1e-12
    assert abs(M.compute_ece(e["conf"], e["ok"], n_bins=10) - e["value10"]) < 1e-12
    assert M.compute_ece([], []) == 0.0


def test_token_metrics_match_reference():
    t = GOLD["token_metrics"]
    logits = torch.tensor(t["logits"]); tgt = torch.tensor(t["targets"])
    nll, lp, ent, conf, ok = M.compute_token_metrics(logits, tgt, logits.argmax(-1))
    assert abs(nll - t["nll"]) < 1e-6 and abs(lp - t["avg_log_prob"]) < 1e-6 and abs(ent - t["entropy"]) < 1e-6
    np.testing.assert_allclose(conf, t["confidences"], rtol=1e-6)
    assert ok == t["correct"]
    assert list(M.compute_token_metrics(logits, torch.full((9,), -100), logits.argmax(-1))) == GOLD["token_metrics_all_pad"]
    # closed forms used by the fused kernel path
    lg = logits.double()
    lse = torch.logsumexp(lg, -1); mx = lg.max(-1).values; ex = (torch.softmax(lg, -1) * lg).sum(-1)
    xt = lg.gather(1, tgt.clamp(min=0).unsqueeze(1)).squeeze(1)
    stats = torch.stack([lse, mx, ex, xt], 1).numpy()
    got = M.token_metrics_from_stats(stats, logits.argmax(-1).numpy(), tgt.numpy())
    assert abs(got[0] - t["nll"]) < 1e-6 and abs(got[1] - t["avg_log_prob"]) < 1e-6 and abs(got[2] - t["entropy"]) < 1e-5
    np.testing.assert_allclose(got[3], t["confidences"], rtol=1e-5)
    assert got[4] == t["correct"]


def test_aggregation_and_macro_average():
    u = [M.PerUtteranceMetrics("a", "a", 0.0, 0.0, 1.0, -1.0, 0.5, [0.9, 0.8], [True, False]),
         M.PerUtteranceMetrics("b", "c", 1.0, 1.0, 3.0, -3.0, 1.5, [0.4], [False])]
    d = M.aggregate_dataset_metrics(u, "ds")
    assert d.num_samples == 2 and d.wer == 0.5 and d.mean_token_nll == 2.0 and d.mean_token_entropy == 1.0
    assert abs(d.ece - M.compute_ece([0.9, 0.8, 0.4], [True, False, False])) < 1e-12
    empty = M.aggregate_dataset_metrics([], "none")
    assert empty.num_samples == 0 and empty.wer == 0.0
    macro = M.compute_macro_average([d, empty])
    assert macro["macro_wer"] == 0.25 and set(macro) == {"macro_wer", "macro_cer", "macro_mean_token_nll", "macro_avg_log_prob",
                                                       "macro_mean_token_entropy", "macro_ece"}
    assert M.compute_macro_average([])["macro_wer"] == 0.0


class _Tok:
    """Duck-typed tokenizer: ids 0..25 -> letters, 26 -> blank; specials 90.."""
    special_tokens = {"<|sot|>": 90, "<|eot|>": 91}

    def decode(self, ids):
        return "".join(" " if i == 26 else chr(97 + i % 26) for i in ids)


@pytest.mark.gpu
def test_token_stats_kernel_matches_reference_metrics():
    from whisper_finetune.engine import kernels as K
    t = GOLD["token_metrics"]
    logits = torch.tensor(t["logits"]).to(torch.bfloat16)
    pad = torch.zeros(9, 128, dtype=torch.bfloat16); pad[:, :50] = logits
    tgt = torch.tensor(t["targets"])
    stats, am = K.token_stats(pad.cuda(), tgt.cuda(), 50)
    assert torch.equal(am.cpu(), logits.float().argmax(-1))  # bit-exact token ids
    ref = M.compute_token_metrics(logits.float(), tgt, logits.float().argmax(-1))
    got = M.token_metrics_from_stats(stats.cpu().numpy(), am.cpu().numpy(), tgt.numpy())
    for a, b in zip(got[:3], ref[:3]):
        assert abs(a - b) < 2e-5
    np.testing.assert_allclose(got[3], ref[3], rtol=1e-4)
    assert got[4] == ref[4]
    # BASELINE-size vocabulary, every row
    g = torch.Generator(device="cuda").manual_seed(0)
    big = (torch.randn(64, 51968, device="cuda", generator=g) * 2).to(torch.bfloat16)
    tg = torch.randint(0, 51866, (64,), device="cuda", generator=g)
    stats, am = K.token_stats(big, tg, 51866)
    lg = big[:, :51866].float()
    assert torch.equal(am, lg.argmax(-1))
    torch.testing.assert_close(stats[:, 0], torch.logsumexp(lg, -1), atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(stats[:, 2], (torch.softmax(lg, -1) * lg).sum(-1), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(stats[:, 3], lg.gather(1, tg[:, None])[:, 0], atol=0, rtol=0)


@pytest.mark.gpu
def test_evaluator_end_to_end_fused_equals_unfused():
    from oracle import whisper_oracle as O
    from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
    from whisper_finetune.eval import evaluator
    dims = ModelDimensions(80, 100, 128, 2, 2, 128, 32, 128, 2, 2)
    params = O.init_params(O.ModelDimensions(**vars(dims)), seed=1, std=0.3)
    m = Whisper(dims); m.load_state_dict(params); m.cuda()
    g = torch.Generator().manual_seed(0)
    batches = []
    for _ in range(2):
        y_out = torch.randint(0, 27, (3, 10), generator=g); y_out[0, -3:] = -100; y_out[2, :] = -100
        batches.append((torch.randn(3, 80, 200, generator=g), torch.randint(0, 27, (3, 10), generator=g), y_out))
    cfg = {"mixed_precision_training": True, "mp_dtype": "bf16"}
    fused = evaluator.evaluate_single_dataset(m, batches, "syn", cfg, tokenizer=_Tok())
    assert fused.num_samples == 4  # the all -100 rows decode to empty references and are skipped

    class Plain(torch.nn.Module):  # hides forward_loss / padded_logits: takes the reference's unfused route
        def __init__(self, inner): super().__init__(); self.inner = inner
        def forward(self, x, y): return self.inner(x, y)
    plain = evaluator.evaluate_single_dataset(Plain(m), batches, "syn", cfg, tokenizer=_Tok())
    assert fused.wer == plain.wer and fused.cer == plain.cer
    assert abs(fused.mean_token_nll - plain.mean_token_nll) < 1e-4
    assert abs(fused.mean_token_entropy - plain.mean_token_entropy) < 1e-4
    assert abs(fused.ece - plain.ece) < 1e-5
    res, macro = evaluator.evaluate_multiple_datasets(m, {"a": batches, "b": batches[:1]}, cfg, tokenizer=_Tok())
    assert set(res) == {"a", "b"} and abs(macro["macro_wer"] - (res["a"].wer + res["b"].wer) / 2) < 1e-12
