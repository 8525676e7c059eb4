"""Launch mode is a PER-CALL argument: `runtime.exchange_launch_mode` is thread-local, the autograd nodes note it at
forward time and their backward kernels carry `launch_mode = 1` in wft_gemm_args / wft_attn_args; libwft holds no mutable launch
state, so an evaluator on another thread and stream keeps its persistent grids while a training backward runs per tile.
Reference: the DDP wrap at scripts/finetune.py:694-710 and the `no_sync()` window of model/model_utils.py:63-72."""
import ctypes as C
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402
from whisper_finetune import runtime as rt  # noqa: E402
from whisper_finetune.engine import kernels as K  # noqa: E402
from whisper_finetune.engine import lib as L  # noqa: E402
from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper  # noqa: E402

DEV = torch.device("cuda:0")


def _model(seed):
    dims = O.DIMS["tiny"]
    m = Whisper(ModelDimensions(**vars(dims)))
    m.load_state_dict(O.init_params(dims, seed=seed))
    return m.to(DEV), dims


def test_backward_beside_an_exchange_runs_per_tile_while_another_thread_stays_persistent(monkeypatch):
    lib = L.load()
    seen = []  # (thread name, entry point, launch_mode)
    lock = threading.Lock()

    def spy(name):
        real = getattr(lib, name)

        def call(args_ref, stream):
            with lock:
                seen.append((threading.current_thread().name, name, int(args_ref._obj.launch_mode)))
            return real(args_ref, stream)

        monkeypatch.setattr(lib, name, call, raising=False)

    for n in ("wft_gemm_nt_bf16", "wft_attn_bwd_bf16", "wft_attn_fwd_bf16"):
        spy(n)

    train, dims = _model(1)
    evalm, _ = _model(2)
    train.train(); evalm.eval()
    audio, y_in, y_out = O.synthetic_batch(dims, 2, 16)
    mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
    y_in, y_out = y_in.to(DEV), y_out.to(DEV)
    with torch.no_grad():
        want = evalm(mel, y_in).float().clone()  # (also builds the evaluator's shadows outside the threads)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        ref_loss = train(mel, y_in, targets=y_out, label_smoothing=0.1)
    ref_loss.backward()
    ref_grads = {n: p.grad.clone() for n, p in train.named_parameters() if p.grad is not None}
    train.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    seen.clear()

    go, done = threading.Event(), threading.Event()
    got = {}

    def evaluator():
        side = torch.cuda.Stream(device=DEV)
        go.wait()
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                got["logits"] = evalm(mel, y_in).float()
        side.synchronize()
        done.set()

    th = threading.Thread(target=evaluator, name="evaluator")
    th.start()
    with rt.exchange_launch_mode(True):  # what train_step enters for the last micro-batch of a window in a multi-process job
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            loss = train(mel, y_in, targets=y_out, label_smoothing=0.1)
        go.set()  # the evaluator's forwards overlap the training backward
        loss.backward()
    done.wait(60)
    th.join(60)
    torch.cuda.synchronize()
    assert rt.backward_launch_mode() == 0

    ev = [s for s in seen if s[0] == "evaluator"]
    main_fwd = [s for s in seen if s[0] == "MainThread"]
    bwd = [s for s in seen if s[0] not in ("evaluator", "MainThread")]  # the autograd engine's device thread
    assert len(ev) > 30 and all(s[2] == 0 for s in ev), "another thread's calls keep the default launch mode"
    assert len(main_fwd) > 30 and all(s[2] == 0 for s in main_fwd), "the forward pass keeps its persistent grids"
    assert len(bwd) > 30 and all(s[2] == 1 for s in bwd), "every backward GEMM / attention call of the traced graph carries launch_mode = 1"
    assert any(s[1] == "wft_attn_bwd_bf16" for s in bwd)
    # same results either way (launch mode never changes arithmetic) — and the evaluator was not disturbed
    assert loss.item() == ref_loss.item()
    for n, p in train.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, ref_grads[n]), (n, float((p.grad - ref_grads[n]).abs().max()), float(ref_grads[n].abs().max()))
    assert torch.equal(got["logits"], want)
