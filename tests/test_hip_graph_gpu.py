"""`training.wft_hip_graph: true`: model_utils.train_step runs each micro-batch (forward + fused loss + backward) as ONE captured HIP
graph (engine/graph.py) — for the launch-bound small configurations of the reference (BASELINE.json configs[1]; loop at
scripts/finetune.py:177-188, train_step at model/model_utils.py:23-127).  Same kernels in the same order as the eager path: losses and
parameters must agree bit for bit; configurations whose kernel arguments are drawn on the host per call are refused loudly."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402
from whisper_finetune.engine import graph as G  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper  # noqa: E402
from whisper_finetune.model import model_utils  # noqa: E402
from whisper_finetune.model.optimizer import WftAdamW  # noqa: E402

DEV = torch.device("cuda:0")


def _run(graph: bool, accum: int, steps: int, name="tiny", save_to=None, drop_grads_after=None):
    dims = O.DIMS[name]
    params = O.init_params(dims, seed=4)
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(params)
    m.to(DEV)
    opt = WftAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))  # a learning rate that changes every step
    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": accum, "max_grad_norm": 1.0, "mp_dtype": "bf16",
             "label_smoothing": 0.1, "wft_hip_graph": graph}
    mels, toks = [], []
    for S in (16, 24):  # two decoder lengths: two graphs sharing the gradient buffers
        audio, y_in, y_out = O.synthetic_batch(dims, 3, S)
        y_out[0, :2] = -100
        mels.append(K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV)))
        toks.append((y_in.to(DEV), y_out.to(DEV)))

    def batches():
        i = 0
        while True:
            j = (i // 3) % 2  # shape changes every third micro-batch
            g = torch.Generator(device=DEV).manual_seed(i)
            yield mels[j] + 0.01 * torch.randn(mels[j].shape, device=DEV, generator=g), toks[j][0], toks[j][1]
            i += 1

    it = batches()
    losses = []
    for i in range(steps):
        losses.append(model_utils.train_step(m, it, opt, sched, t_cfg))
        if drop_grads_after is not None and i == drop_grads_after:
            opt.zero_grad(set_to_none=True)  # (what an evaluation pass between training steps may do)
    gm = G.graphed_for(m)
    if graph and save_to is not None:
        # the reference saves checkpoints from inside the loop (scripts/finetune.py:205-226 -> model_utils.save_model: deepcopy of the
        # model): must work once graphs are captured (the graph cache used to live in the module's __dict__)
        model_utils.save_model(m, save_to)
        ck = torch.load(save_to, map_location="cpu")
        for n, p in m.named_parameters():
            assert torch.equal(ck["model_state_dict"][n], p.detach().half().cpu()), n
        losses += [model_utils.train_step(m, it, opt, sched, t_cfg)]  # and the graphs keep working afterwards
    return losses, {n: p.detach().clone() for n, p in m.named_parameters()}, (len(gm[1].graphs) if gm else 0)


@pytest.mark.parametrize("accum", [1, 3])
def test_graphed_steps_equal_eager_steps_bit_for_bit(accum):
    l0, p0, n0 = _run(False, accum, 6)
    l1, p1, n1 = _run(True, accum, 6)
    assert n0 == 0 and n1 == 2          # one graph per input shape, captured behind the eager warm-up micro-batches
    assert l0 == l1, (l0, l1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n


def test_save_model_after_capture(tmp_path):
    import copy

    losses, params, n = _run(True, 1, 5, save_to=str(tmp_path / "ck.pt"))
    assert n == 2 and len(losses) == 6 and all(l > 0 for l in losses)


def test_shape_cap_falls_back_to_eager(capsys, monkeypatch):
    monkeypatch.setattr(G, "MAX_SHAPES", 1)
    l0, p0, n0 = _run(False, 1, 6)
    l1, p1, n1 = _run(True, 1, 6)
    assert n1 == 1 and l0 == l1
    assert "further shapes run eagerly" in capsys.readouterr().out
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n


def test_configurations_with_host_drawn_kernel_arguments_are_refused(capsys):
    from whisper_finetune.model import lora
    from whisper_finetune.model.model_utils import CheckpointedStochasticTextDecoder, register_deep_spec_augment_hooks

    dims = O.DIMS["tiny"]
    m = Whisper(ModelDimensions(**vars(dims))).to(DEV)
    assert G.why_not(m) is None
    m.decoder = CheckpointedStochasticTextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer, 0.1).to(DEV)
    assert "stochastic depth" in G.why_not(m)
    m2 = Whisper(ModelDimensions(**vars(dims)))
    m2.load_state_dict(O.init_params(dims, seed=4))
    m2.to(DEV)
    lora.apply_lora(m2, {"rank": 4, "lora_alpha": 8, "lora_dropout": 0.1})
    assert "LoRA dropout" in G.why_not(m2)
    m3 = Whisper(ModelDimensions(**vars(dims))).to(DEV)
    register_deep_spec_augment_hooks(m3, 100, 27)
    assert "deep SpecAugment" in G.why_not(m3)
    assert "HIP device" in G.why_not(Whisper(ModelDimensions(**vars(dims))))
    # train_step says so once and runs the eager path
    opt = WftAdamW(m2.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 8)
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))

    def it():
        while True:
            yield mel, y_in.to(DEV), y_out.to(DEV)

    t_cfg = {"mixed_precision_training": True, "accum_grad_steps": 1, "max_grad_norm": 1.0, "mp_dtype": "bf16", "wft_hip_graph": True,
             "is_lora_run": False}
    gen = it()
    for _ in range(2):
        assert model_utils.train_step(m2, gen, opt, sched, t_cfg) > 0
    out = capsys.readouterr().out
    assert out.count("stays on the eager path") == 1 and not G.has_graphs(m2)


def test_gradients_dropped_between_steps_come_back_as_their_persistent_slices():
    """Round 6: the graph's gradients are slices of one flat buffer, zeroed with one fill.  `zero_grad(set_to_none=True)` from outside
    (an evaluation loop) must not detach the optimizer from them."""
    l0, p0, _ = _run(False, 2, 7, drop_grads_after=4)
    l1, p1, n1 = _run(True, 2, 7, drop_grads_after=4)
    assert n1 == 2 and l0 == l1, (l0, l1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n
