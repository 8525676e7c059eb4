"""Developer script (GPU box): where the wall time of tests/test_large_v3_gpu.py goes (phase timings of the full-fine-tune case)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "whisper-finetune_amd"))
import torch
t0 = time.time()
def lap(tag):
    global t0
    torch.cuda.synchronize(); t = time.time(); print(f"{tag:40s} {t - t0:7.1f} s", flush=True); t0 = t
import tests.test_large_v3_gpu as T
from oracle import whisper_oracle as O
from whisper_finetune.engine import kernels as K
from whisper_finetune.engine.whisper_model import ModelDimensions, Whisper
DEV = torch.device("cuda:0")
lap("imports")
dims, params = T._large_v3_params(); lap("_large_v3_params (init + biases)")
audio, y_in, y_out = O.synthetic_batch(dims, 1, 128); mel_ref = O.log_mel_spectrogram(audio, dims.n_mels); lap("synthetic batch + oracle log-mel")
m = Whisper(ModelDimensions(**vars(dims))); lap("Whisper(dims) construction")
m.load_state_dict(params); lap("load_state_dict")
m.to(DEV).train(); lap("to(cuda)")
mel = K.logmel(audio.to(DEV), O.mel_filters(dims.n_mels).to(DEV))
loss = m(mel, y_in.to(DEV), targets=y_out.to(DEV), label_smoothing=0.1); loss.backward(); lap("engine forward + backward")
got = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}; lap("gradients to host")
mel_cpu = mel.cpu()
def run(emulate):
    p_req = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in params.items()}
    T._oracle_grads(dims, p_req, mel_cpu if emulate else mel_ref, y_in, y_out, emulate)
    return {n: p_req[n].grad for n in got}
refs = T._both_oracles(run); lap("both oracle passes, two host threads")
T._check(got, refs[True], refs[False], "full-FT"); lap("_check (3 x 1259 relative errors)")
print("threads", torch.get_num_threads())
