"""Generates the golden fixtures in tests/golden/*.npz.  Run ONLY in the build container
(it reads /root/reference and uses HF transformers); the fixtures are committed, this script
is their provenance.

    python tests/golden/gen_golden.py

Sources of truth
  whisper_arch.npz   HF transformers WhisperForConditionalGeneration (independent implementation
                     of the architecture) loaded with oracle-named weights through the REFERENCE's
                     key map (scripts/convert_openai_to_hf.py:89-110): logits, loss, gradients.
  logmel.npz         HF WhisperFeatureExtractor (numpy STFT, slaney filterbank) on seeded clips,
                     and transformers.audio_utils.mel_filter_bank for the filterbank itself.
  ref_host.npz       the REFERENCE'S OWN Python (imported with import stubs for the third-party
                     packages that are not installed): TimeWarpAugmenter, ExtremesFrequencyMasking,
                     pad_or_trim, StochasticDepthMixin, register_deep_spec_augment_hooks,
                     calculate_training_steps / val_steps / resolve_local_accum_grad_steps.
"""
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
from oracle import whisper_oracle as O  # noqa: E402

REF = Path("/root/reference")

# small dims that still satisfy the kernels' constraints (d % 128 == 0, head_dim 64)
ARCH_DIMS = O.ModelDimensions(n_mels=80, n_audio_ctx=100, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                              n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)


def arch_params(dims, seed):
    p = O.init_params(dims, seed=seed, std=0.08)
    g = torch.Generator().manual_seed(seed + 100)
    for k, v in p.items():
        if k.endswith(".bias"):
            p[k] = torch.randn(v.shape, generator=g) * 0.05
        elif "ln" in k and k.endswith(".weight"):
            p[k] = 1 + torch.randn(v.shape, generator=g) * 0.1
    return p


def arch_inputs(dims, seed):
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(2, dims.n_mels, 2 * dims.n_audio_ctx, generator=g)
    y_in = torch.randint(0, dims.n_vocab, (2, 12), generator=g)
    y_out = torch.randint(0, dims.n_vocab, (2, 12), generator=g)
    y_out[0, :3] = -100
    y_out[1, -2:] = -100
    return mel, y_in, y_out


def gen_arch():
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    sys.path.insert(0, str(REF / "src"))
    src = (REF / "src/whisper_finetune/scripts/convert_openai_to_hf.py").read_text()
    ns = {}
    start = src.index("WHISPER_MAPPING = {")
    exec(src[start: src.index("}", start) + 1], ns)  # only the key map literal
    mapping = ns["WHISPER_MAPPING"]

    dims = ARCH_DIMS
    params = arch_params(dims, 3)
    hf_sd = {}
    for k, v in params.items():
        nk = k
        for a, b in mapping.items():
            if a in nk:
                nk = nk.replace(a, b)
        hf_sd[nk] = v.clone()
    cfg = WhisperConfig(vocab_size=dims.n_vocab, encoder_ffn_dim=4 * dims.n_audio_state, decoder_ffn_dim=4 * dims.n_text_state,
                        num_mel_bins=dims.n_mels, d_model=dims.n_audio_state, max_target_positions=dims.n_text_ctx,
                        encoder_layers=dims.n_audio_layer, encoder_attention_heads=dims.n_audio_head,
                        decoder_layers=dims.n_text_layer, decoder_attention_heads=dims.n_text_head,
                        max_source_positions=dims.n_audio_ctx, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                        eos_token_id=1, bos_token_id=1, pad_token_id=1, decoder_start_token_id=2, attn_implementation="eager")
    model = WhisperForConditionalGeneration(cfg)
    missing, unexpected = model.model.load_state_dict(hf_sd, strict=False)
    assert not unexpected, unexpected
    assert all("embed_positions" in m for m in missing), missing
    model.proj_out.weight = model.model.decoder.embed_tokens.weight  # tied (convert_openai_to_hf.py:223-224)
    model.train()
    mel, y_in, y_out = arch_inputs(dims, 11)
    logits = model(input_features=mel, decoder_input_ids=y_in).logits
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing=0.1)
    loss.backward()
    inv = {}
    for k in params:
        nk = k
        for a, b in mapping.items():
            if a in nk:
                nk = nk.replace(a, b)
        inv[k] = nk
    named = dict(model.model.named_parameters())
    out = {"logits": logits.detach().numpy(), "loss": np.float64(loss.item())}
    norms = {}
    for k, nk in inv.items():
        if nk in named and named[nk].grad is not None:
            norms[k] = named[nk].grad.norm().item()
    out["grad_norm_names"] = np.array(sorted(norms))
    out["grad_norms"] = np.array([norms[k] for k in sorted(norms)])
    for k in ["encoder.conv1.weight", "encoder.blocks.0.attn.key.weight", "encoder.blocks.1.mlp.0.bias",
              "decoder.blocks.1.cross_attn.query.weight", "decoder.blocks.0.attn_ln.weight", "decoder.positional_embedding"]:
        out["grad::" + k] = named[inv[k]].grad.numpy()
    np.savez_compressed(HERE / "whisper_arch.npz", **out)
    print("whisper_arch.npz: loss", loss.item(), "params with grads:", len(norms))


def gen_logmel():
    from transformers import WhisperFeatureExtractor
    from transformers.audio_utils import mel_filter_bank

    out = {}
    clips = []
    for i in range(2):
        g = torch.Generator().manual_seed(1234 + i)
        a = torch.randn(O.N_SAMPLES, generator=g) * 0.1
        if i == 1:
            a[200000:] = 0.0  # zero-padded tail (data_loader.py:346)
            a[:200000] *= torch.linspace(0.0, 1.0, 200000)
        clips.append(a.numpy())
    t = np.arange(O.N_SAMPLES) / 16000.0
    clips.append((0.3 * np.sin(2 * np.pi * (200.0 + 120.0 * t) * t)).astype(np.float32))  # sine sweep
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        feats = fe(clips, sampling_rate=16000, return_tensors="np")["input_features"]  # [3, n_mels, 3000]
        out[f"mel{n_mels}_sub"] = feats[:, :, ::7].astype(np.float32)
        out[f"filters{n_mels}"] = mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T.astype(np.float32)
    np.savez_compressed(HERE / "logmel.npz", **out)
    print("logmel.npz written")


def _install_stubs(tmp: Path):
    (tmp / "whisper").mkdir()
    (tmp / "whisper/__init__.py").write_text(
        "class Whisper: pass\n_ALIGNMENT_HEADS={}\n_MODELS={}\ndef _download(*a,**k): raise RuntimeError\n"
        "def available_models(): return []\ndef load_model(*a,**k): raise RuntimeError\n")
    (tmp / "whisper/audio.py").write_text(
        "SAMPLE_RATE=16000\nN_FFT=400\nHOP_LENGTH=160\nCHUNK_LENGTH=30\nN_SAMPLES=480000\nN_FRAMES=3000\n"
        "def log_mel_spectrogram(*a,**k): raise RuntimeError\n")
    (tmp / "whisper/tokenizer.py").write_text("LANGUAGES={}\nTO_LANGUAGE_CODE={}\nclass Tokenizer: pass\ndef get_tokenizer(*a,**k): raise RuntimeError\n")
    (tmp / "whisper/model.py").write_text(
        "import torch\nclass Linear(torch.nn.Linear): pass\nclass AudioEncoder(torch.nn.Module):\n    def __init__(self,*a,**k): super().__init__()\n"
        "class TextDecoder(torch.nn.Module):\n    def __init__(self,*a,**k): super().__init__()\nclass Whisper(torch.nn.Module): pass\n")
    (tmp / "torchaudio").mkdir()
    (tmp / "torchaudio/__init__.py").write_text("")
    # torchaudio.transforms restated from the published mask_along_axis
    (tmp / "torchaudio/transforms.py").write_text(
        "import torch\n"
        "class _M(torch.nn.Module):\n"
        "    def __init__(self, p, axis):\n        super().__init__(); self.p=p; self.axis=axis\n"
        "    def forward(self, x):\n"
        "        size=x.size(self.axis)\n        value=torch.rand(1)*self.p\n        mn=torch.rand(1)*(size-value)\n"
        "        s=int(mn.long()); e=s+int(value.long())\n        x=x.clone()\n"
        "        idx=[slice(None)]*x.dim(); idx[self.axis]=slice(s,e); x[tuple(idx)]=0.0\n        return x\n"
        "class TimeMasking(_M):\n    def __init__(self, time_mask_param): super().__init__(time_mask_param, -1)\n"
        "class FrequencyMasking(_M):\n    def __init__(self, freq_mask_param): super().__init__(freq_mask_param, -2)\n")
    (tmp / "audiomentations.py").write_text("def __getattr__(name):\n    return type(name, (), {'__init__': lambda self, *a, **k: None})\n")
    (tmp / "jiwer.py").write_text("def wer(*a,**k): raise RuntimeError\ndef cer(*a,**k): raise RuntimeError\n")


def gen_ref_host():
    tmp = Path(tempfile.mkdtemp())
    _install_stubs(tmp)
    sys.path.insert(0, str(tmp))
    sys.path.insert(0, str(REF / "src"))
    sys.dont_write_bytecode = True
    from whisper_finetune import utils as rutils
    from whisper_finetune.data import utils as rdata
    from whisper_finetune.model import model_utils as rmu

    out = {}
    # --- time warp (data/utils.py:41-143) on a smooth + noisy spectrogram, several draws
    g = torch.Generator().manual_seed(7)
    spec = torch.randn(16, 300, generator=g).cumsum(1) * 0.1
    out["tw_spec"] = spec.numpy()
    for i, seed in enumerate((0, 1, 2)):
        torch.manual_seed(seed)
        W = 20
        # replay the two draws the reference makes to record them
        st = torch.get_rng_state()
        wp = int(torch.randint(W, 300 - W, (1,)))
        wd = int(torch.randint(-W, W, (1,)))
        torch.set_rng_state(st)
        warped = rdata.TimeWarpAugmenter(W=W)(spec.clone())
        out[f"tw_params{i}"] = np.array([wp, wd])
        out[f"tw_out{i}"] = warped.numpy()
    # --- extremes frequency masking (data/utils.py:146-190)
    torch.manual_seed(5)
    st = torch.get_rng_state()
    r = torch.rand(1).item()
    torch.set_rng_state(st)
    ex = rdata.ExtremesFrequencyMasking(low_freq_range=6, high_freq_range=4)(torch.ones(16, 10))
    out["ext_r"] = np.float64(r)
    out["ext_out"] = ex.numpy()
    # --- pad_or_trim with the minimum (data/utils.py:380-394)
    short = torch.arange(12, dtype=torch.float32).reshape(3, 4) - 5
    out["pad_in"] = short.numpy()
    out["pad_out"] = rdata.pad_or_trim(short, 7).numpy()
    out["trim_out"] = rdata.pad_or_trim(short, 2).numpy()
    # --- stochastic depth arithmetic (model/model_utils.py:226-250)
    class SD(rmu.StochasticDepthMixin, torch.nn.Module):
        pass
    sd = SD(); sd.train()
    x = torch.tensor([[1.0, -2.0, 3.0]])
    torch.manual_seed(0)
    draws = []
    outs = []
    for _ in range(6):
        st = torch.get_rng_state(); d = torch.rand(1).item(); torch.set_rng_state(st)
        draws.append(d)
        outs.append(sd.stochastic_depth(x, lambda t: t * 2 + 1, 0.4).detach().numpy())
    out["sd_draws"] = np.array(draws)
    out["sd_outs"] = np.stack(outs)
    sd.eval()
    out["sd_eval"] = sd.stochastic_depth(x, lambda t: t * 2 + 1, 0.4).detach().numpy()
    # --- deep SpecAugment hooks (model/model_utils.py:382-437): which rows / channels get zeroed
    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.attn_ln = torch.nn.Identity()
        def forward(self, x): return self.attn_ln(x)
    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.blocks = torch.nn.ModuleList([Blk() for _ in range(3)])
        def forward(self, x):
            outs = []
            for b in self.blocks: outs.append(b(x))
            return outs
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.encoder = Enc()
    m = M(); m.train()
    rmu.register_deep_spec_augment_hooks(m, time_mask_param=30, freq_mask_param=20, p=1.0)
    torch.manual_seed(42)
    ys = m.encoder(torch.ones(2, 150, 128))
    for i, y in enumerate(ys):
        z = (y == 0)
        out[f"dsa_rows{i}"] = z.all(dim=2)[0].numpy()    # [T] time steps fully zeroed
        out[f"dsa_cols{i}"] = z.all(dim=1)[0].numpy()    # [d] channels fully zeroed
    # --- step arithmetic tables (utils.py:14-53)
    rows = []
    for n in (1000, 1234, 99):
        for world in (1, 2, 4, 8):
            for bs in (4, 32):
                for ep in (1, 2.5):
                    for acc in (1, 4):
                        for dl in (True, False):
                            cfg = {"training": {"epochs": ep, "accum_grad_steps": acc}, "dataset": {"batch_size": bs}}
                            rows.append([n, world, bs, ep, acc, int(dl), rutils.calculate_training_steps(cfg, range(n), world, dl)])
    out["train_steps_table"] = np.array(rows, dtype=np.float64)
    vs = []
    for ts in (10, 137, 1000):
        for ep in (1, 3):
            for ev in (0.1, 0.25, 1.0):
                vs.append([ts, ep, ev, rutils.calculate_val_steps({"training": {"train_steps": ts, "epochs": ep, "eval_steps": ev}})])
    out["val_steps_table"] = np.array(vs, dtype=np.float64)
    out["local_accum"] = np.array([[a, w, rutils.resolve_local_accum_grad_steps(a, w)] for a, w in ((8, 1), (8, 2), (8, 4), (8, 8), (4, 2))])
    np.savez_compressed(HERE / "ref_host.npz", **out)
    print("ref_host.npz written", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_ref_eval():
    """eval/utils.py normaliser, eval/metrics.py ECE / token metrics / aggregation from the reference's own code."""
    import json
    tmp = Path(tempfile.mkdtemp())
    _install_stubs(tmp)
    sys.path.insert(0, str(tmp))
    sys.path.insert(0, str(REF / "src"))
    from whisper_finetune.eval import metrics as rm
    from whisper_finetune.eval import utils as ru

    out = {}
    texts = ["Grüß Gott, wie geht's? Über-Straße 12/3 – ÇA VA", "  Hello\tWORLD  ñandú şah Ș ", "a-b–c/d ÄÖÜ äöü ß",
             "1,234.5 test: ok; yes!", "", "   "]
    out["normalize"] = {v: [[t, ru.normalize_text(t, **ru.VOCAB_SPECS[v])] for t in texts] for v in ("v0", "v1", "v2", "v3")}
    g = torch.Generator().manual_seed(3)
    conf = torch.rand(500, generator=g).tolist()
    ok = (torch.rand(500, generator=g) < torch.tensor(conf)).tolist()
    out["ece"] = {"conf": conf, "ok": ok, "value": float(rm.compute_ece(conf, ok)), "value10": float(rm.compute_ece(conf, ok, n_bins=10))}
    logits = torch.randn(9, 50, generator=g) * 2
    tgt = torch.randint(0, 50, (9,), generator=g); tgt[[2, 7]] = -100
    pred = logits.argmax(-1)
    nll, lp, ent, cf, cr = rm.compute_token_metrics(logits, tgt, pred)
    out["token_metrics"] = {"logits": logits.tolist(), "targets": tgt.tolist(), "nll": nll, "avg_log_prob": lp, "entropy": ent,
                            "confidences": cf, "correct": cr}
    nll0 = rm.compute_token_metrics(logits, torch.full((9,), -100), pred)
    out["token_metrics_all_pad"] = list(nll0)
    (HERE / "ref_eval.json").write_text(json.dumps(out, indent=0))
    print("ref_eval.json written")


def fake_muon_model():
    """A whisper-shaped toy (encoder/decoder block stacks with 2-D Linear weights of two widths, biases, norms; a conv
    stem, embeddings and a final norm outside the blocks) for the optimizer-factory parity case."""
    torch.manual_seed(0)

    class Block(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.attn_ln = torch.nn.LayerNorm(d)
            self.query = torch.nn.Linear(d, d)
            self.key = torch.nn.Linear(d, d, bias=False)
            self.mlp = torch.nn.Sequential(torch.nn.Linear(d, 4 * d), torch.nn.GELU(), torch.nn.Linear(4 * d, d))

    class Stack(torch.nn.Module):
        def __init__(self, d, n, conv):
            super().__init__()
            if conv:
                self.conv1 = torch.nn.Conv1d(8, d, 3, padding=1)
            else:
                self.token_embedding = torch.nn.Embedding(32, d)
            self.blocks = torch.nn.ModuleList([Block(d) for _ in range(n)])
            self.ln_post = torch.nn.LayerNorm(d)

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Stack(16, 2, True)
            self.decoder = Stack(16, 1, False)

    m = Fake()
    m.encoder.blocks[0].key.weight.requires_grad = False  # frozen parameters are left out of every group
    return m


MUON_CONF = {"type": "adamw", "muon": True, "8bit": False, "muon_ndim_threshold": 2,
             "muon_params": {"lr": 2e-4, "momentum": 0.95, "weight_decay": 0.01},
             "params": {"lr": 2e-5, "weight_decay": 0.01, "betas": [0.9, 0.98], "eps": 1e-6, "amsgrad": False}}


def gen_ref_optim():
    """Param-group construction of the reference's get_optimizer Muon branch (model/optimizer.py:9-128), from the reference's
    own helper functions (the `muon` package itself is not installed, so the optimizer object is not built)."""
    import json
    tmp = Path(tempfile.mkdtemp())
    _install_stubs(tmp)
    sys.path.insert(0, str(tmp))
    sys.path.insert(0, str(REF / "src"))
    from whisper_finetune.model import optimizer as ro

    m = fake_muon_model()
    names = {id(p): n for n, p in m.named_parameters()}
    muon, aux = ro._partition_muon_params(m, ndim_threshold=2)
    out = {"muon_names": [names[id(p)] for p in muon], "aux_names": [names[id(p)] for p in aux], "groups": {}}
    for match in (True, False):
        groups = ro._build_muon_param_groups(muon, base_lr=2e-4, base_weight_decay=0.01, momentum=0.95,
                                             match_adamw_update_rms=match, match_factor=0.2)
        out["groups"][str(match)] = [{"names": [names[id(p)] for p in g["params"]], "lr": g["lr"], "momentum": g["momentum"],
                                      "weight_decay": g["weight_decay"], "use_muon": g["use_muon"]} for g in groups]
    out["use_muon"] = [[c, ro._use_muon_optimizer(c)] for c in ({"muon": True}, {"muon": False, "type": "muon"}, {"type": "muon"}, {"type": "adamw"})]
    (HERE / "ref_optim.json").write_text(json.dumps(out, indent=0))
    print("ref_optim.json written", len(muon), len(aux))


def _ref_imports():
    """sys.path set up so that `whisper_finetune` is the REFERENCE's package (with the import stubs)."""
    tmp = Path(tempfile.mkdtemp())
    _install_stubs(tmp)
    sys.path.insert(0, str(tmp))
    sys.path.insert(0, str(REF / "src"))
    sys.dont_write_bytecode = True
    return tmp


def gen_ref_sched():
    """LR tables from the reference's own get_scheduler (model/scheduler.py:114-151; transformers' schedules for linear /
    cosine / cosine_with_restarts, the in-tree lambdas for the warm-restart variants).  The chill variant draws
    random.uniform per step: the table is taken under random.seed(0), which the test repeats."""
    import json
    import random
    _ref_imports()
    from whisper_finetune.model import scheduler as rs

    confs = {
        "linear": {"type": "linear", "warmup_steps": 10},
        "cosine": {"type": "cosine", "warmup_steps": 7},
        "cosine_with_restarts": {"type": "cosine_with_restarts", "warmup_steps": 5, "lr_num_cycles": 3},
        "cosine_with_warmup_restarts": {"type": "cosine_with_warmup_restarts", "warmup_steps": 6, "lr_num_cycles": 4, "lr_gamma": 0.8},
        "cosine_with_warmup_restarts_chill": {"type": "cosine_with_warmup_restarts_chill", "warmup_steps": 6, "lr_num_cycles": 3,
                                              "lr_gamma": 0.9, "chill_steps": 12, "chill_range": 0.02},
    }
    out = {}
    for kind, conf in confs.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        random.seed(0)
        s = rs.get_scheduler(opt, dict(conf), 120)
        lrs = []
        for _ in range(125):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step(); s.step()
        out[kind] = {"conf": conf, "lrs": lrs}
    (HERE / "ref_sched.json").write_text(json.dumps(out, indent=0))
    print("ref_sched.json written", {k: len(v["lrs"]) for k, v in out.items()})


class OracleModule(torch.nn.Module):
    """nn.Module face of oracle.Oracle (parameters registered under their openai-whisper names, '.' -> '/'), so the
    reference's train_step / optimizers can drive the CPU oracle: model(mel, tokens) -> logits f32 [B, S, V]."""

    def __init__(self, dims, params):
        super().__init__()
        self.dims = dims
        self.names = list(params)
        self.buffers_ = {k: v for k, v in params.items() if k == "encoder.positional_embedding"}
        self.ps = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone()) for k, v in params.items()
                                          if k != "encoder.positional_embedding"})

    def state(self):
        d = {k.replace("/", "."): v for k, v in self.ps.items()}
        d.update(self.buffers_)
        return d

    def forward(self, mel, tokens):
        return O.Oracle(self.dims, self.state()).forward(mel, tokens)


def train_step_case(seed=11, steps=4, accum=2, B=2, S=12):
    """Inputs of the train_step golden: `steps * accum` micro-batches of B synthetic (mel, y_in, y_out)."""
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(steps * accum):
        mel = torch.randn(B, ARCH_DIMS.n_mels, 2 * ARCH_DIMS.n_audio_ctx, generator=g)
        y_in = torch.randint(0, ARCH_DIMS.n_vocab, (B, S), generator=g)
        y_out = torch.randint(0, ARCH_DIMS.n_vocab, (B, S), generator=g)
        y_out[0, :2] = -100
        batches.append((mel, y_in, y_out))
    return batches


TRAIN_STEP_CFG = {"mixed_precision_training": False, "accum_grad_steps": 2, "max_grad_norm": 0.5, "mp_dtype": "bf16",
                  "label_smoothing": 0.1}
TRAIN_STEP_OPT = {"lr": 2e-3, "weight_decay": 0.1, "betas": (0.9, 0.98), "eps": 1e-6}


def gen_ref_train_step():
    """the REFERENCE'S OWN train_step (model/model_utils.py:23-127) drives the CPU oracle model for 4
    optimizer steps x 2 micro-batches (fp32, label smoothing 0.1, clip 0.5, AdamW, linear schedule with warm-up):
    the loss sequence, the learning rates and the norms of every parameter afterwards."""
    _ref_imports()
    import warnings
    from whisper_finetune.model import model_utils as rmu
    from whisper_finetune.model import scheduler as rs

    model = OracleModule(ARCH_DIMS, arch_params(ARCH_DIMS, seed=3))
    opt = torch.optim.AdamW(model.parameters(), **TRAIN_STEP_OPT)
    sched = rs.get_scheduler(opt, {"type": "linear", "warmup_steps": 2}, 4)
    it = iter(train_step_case())
    losses, lrs = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # autocast("cuda", enabled=False) on a CPU-only box
        for step in range(1, 5):
            lrs.append(opt.param_groups[0]["lr"])
            losses.append(rmu.train_step(model, it, opt, sched, dict(TRAIN_STEP_CFG), step=step))
    out = {"losses": np.array(losses, dtype=np.float64), "lrs": np.array(lrs, dtype=np.float64)}
    for k, v in model.state().items():
        out["final_norm/" + k] = np.float64(v.detach().double().norm().item())
    np.savez_compressed(HERE / "ref_train_step.npz", **out)
    print("ref_train_step.npz written", losses)


class GoldenTokenizer:
    """Whisper-shaped stub tokenizer of the data-path golden (byte-level text tokens; the v2 special-token layout)."""
    eot, sot, sot_prev, no_speech, no_timestamps, timestamp_begin = 50257, 50258, 50361, 50362, 50363, 50364
    special_tokens = {"<|de|>": 50261, "<|en|>": 50259, "<|transcribe|>": 50359}

    def encode(self, text, **kwargs):
        return list(text.encode("utf-8"))


class GoldenRecords:
    column_names = ["audio", "text", "language", "prompt"]

    def __init__(self, records):
        self.records = records

    def with_format(self, type=None):
        return self

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        r = self.records[i]
        if isinstance(r, Exception):
            raise r
        return dict(r)


def dataset_cases():
    """(records, [(dataset kwargs, index, seed)]) of the data-path golden: prompts, timestamps, partial-segment cut,
    no-speech, prompt truncation (> max_prompt_length and > 448 total), an unreadable row, SpecAugment draws."""
    def rec(text, prompt="", lang="de", secs=1.0):
        return {"audio": {"array": torch.zeros(int(secs * 16000))}, "text": text, "language": lang, "prompt": prompt}
    records = [
        rec("<|0.00|>hallo welt<|2.40|><|2.40|>noch ein satz<|5.00|>", prompt="<|1.00|>vorher<|3.00|>"),
        rec("<|0.00|>teil eins<|3.00|><|3.00|>", prompt="kontext ohne zeitstempel"),      # partial segment -> cut at 3.00 s
        rec("", prompt="p"),                                                                    # empty text -> no_speech
        rec("<|0.00|>" + "x" * 300 + "<|9.98|>", prompt="y" * 400, lang="en"),                   # > 448: the prompt is shortened
        RuntimeError("corrupt row"),                                                            # skipped lazily
        rec("ohne zeitstempel", prompt="<|0.00|>z<|0.50|>"),
    ]
    sa = {"time_mask_param": 100, "freq_mask_param": 27, "time_warp_w": 80, "p": 1.0}
    ex = {"low_freq_range": 10, "high_freq_range": 6}
    base = dict(no_timestamp_training=False, max_prompt_length=223, prompt_use_rate=1.0, no_timestamps_rate=0.0)
    runs = []
    for idx in range(len(records)):
        runs.append((dict(base), idx, 100 + idx))
        runs.append(({**base, "no_timestamp_training": True}, idx, 200 + idx))
        runs.append(({**base, "prompt_use_rate": 0.0, "no_timestamps_rate": 1.0}, idx, 300 + idx))
        runs.append(({**base, "prompt_use_rate": 0.5, "no_timestamps_rate": 0.5, "spec_augment": True, "spec_augment_params": sa,
                      "extremes_spec_augment": True, "extremes_spec_augment_params": ex}, idx, 400 + idx))
        runs.append(({**base, "prompt_use_rate": 0.5, "no_timestamps_rate": 0.5, "spec_augment": True,
                      "spec_augment_params": {**sa, "p": 0.5}}, idx, 500 + idx))
    runs.append(({**base, "max_prompt_length": 5}, 0, 600))
    return records, runs


def gen_ref_dataset():
    """The REFERENCE'S OWN AudioDataset.__getitem__ (data/data_loader.py:190-359) on stub records / tokenizer: decoder
    input and target sequences, the partial-segment cut (read off a ramp standing in for the log-mel), the positions the
    reference's SpecAugment zeroed, and the next default-generator draw after the item (proves the same NUMBER and ORDER of
    draws)."""
    import json
    tmp = _ref_imports()
    # a log-mel stand-in that makes the cut / masks readable: mel[m, t] = 1 + t + 3000 * m  (all > 0, minimum 1)
    (tmp / "whisper/audio.py").write_text(
        "import torch\nSAMPLE_RATE=16000\nN_FFT=400\nHOP_LENGTH=160\nCHUNK_LENGTH=30\nN_SAMPLES=480000\nN_FRAMES=3000\n"
        "def log_mel_spectrogram(audio, n_mels=80, padding=0, device=None):\n"
        "    return 1.0 + torch.arange(3000.0)[None, :] + 3000.0 * torch.arange(float(n_mels))[:, None]\n")
    for mod in [m for m in sys.modules if m.startswith("whisper")]:
        del sys.modules[mod]
    from whisper_finetune.data import data_loader as rdl

    records, runs = dataset_cases()
    ds_records = GoldenRecords(records)
    out = []
    for kw, idx, seed in runs:
        ds = rdl.AudioDataset(ds_records, GoldenTokenizer(), n_mels=80, **kw)
        torch.manual_seed(seed)
        mel, y_in, y_out = ds[idx]
        nxt = torch.rand(1).item()
        m = mel.numpy()
        ramp = 1.0 + np.arange(3000.0)[None, :] + 3000.0 * np.arange(80.0)[:, None]
        zero_t = np.where((m == 0).all(axis=0))[0]
        zero_f = np.where((m == 0).all(axis=1))[0]
        # frames kept before the minimum-value pad: only readable without a time warp (the non-augmented runs)
        untouched = not kw.get("spec_augment", False)
        kept = int((m[0] == ramp[0]).sum()) if untouched else None
        if untouched and kept < 3000:
            assert (m[:, kept:] == m[:, :kept].min()).all()
        out.append({"kw": kw, "index": idx, "seed": seed, "y_in": y_in.tolist(), "y_out": y_out.tolist(), "kept_frames": kept,
                    "zero_time": [int(zero_t.min()), int(zero_t.max()) + 1] if zero_t.size else None,
                    "zero_mels": sorted(int(v) for v in zero_f), "next_rand": nxt,
                    "invalid": sorted(ds.invalid_indices)})
    bad = []
    ds = rdl.AudioDataset(GoldenRecords([{"audio": {"array": torch.zeros(16000)}, "text": "<|0.01|>odd", "language": "de", "prompt": ""}]),
                          GoldenTokenizer(), n_mels=80, no_timestamps_rate=0.0, prompt_use_rate=0.0)
    try:
        ds[0]
    except Exception as exc:
        bad.append([type(exc).__name__, str(exc)])
    (HERE / "ref_dataset.json").write_text(json.dumps({"runs": out, "errors": bad}, indent=0))
    print("ref_dataset.json written", len(out), bad)



# ------------------------------------------------------------------ §8(f3): checkpoint / merge wire format
# (the reference's converter hard-codes Whisper's special-token ids: the vocabulary must be the real one)
CKPT_DIMS = O.ModelDimensions(n_mels=80, n_audio_ctx=100, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                              n_vocab=51865, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)


def gen_ref_checkpoint_make(outdir=None):
    """Step 1 (THIS package): full.pt and lora.pt of the tiny test model, written by our save_model."""
    sys.path.insert(0, str(ROOT / "tests"))
    import _ckpt_case

    out = Path(outdir or sys.argv[2])
    _ckpt_case.build(out, CKPT_DIMS, arch_params(CKPT_DIMS, 3))
    print("wrote", sorted(p.name for p in out.iterdir()))


def gen_ref_checkpoint_consume(outdir=None):
    """Step 2 (the REFERENCE's package, import stubs for the third-party modules): full.pt through
    scripts/convert_openai_to_hf.py:172-224 (convert_openai_whisper_to_tfms) -> HF logits; lora.pt through the flow of
    scripts/merge_lora_weights.py:26-60 (apply_lora -> load_state_dict -> merge_lora) -> merged weights."""
    out = Path(outdir or sys.argv[2])
    tmp = _ref_imports()
    sys.path.insert(0, str(ROOT / "oracle" / "stubs"))  # minlora (restated, pinned by the reference's tests/test_lora.py)
    import whisper_finetune.scripts.convert_openai_to_hf as conv
    from transformers import GenerationConfig

    # the converter's LAST step downloads openai/whisper-*'s generation_config.json from the hub (no network here): replaced by a
    # default GenerationConfig — it does not touch the weights or the forward this fixture records
    conv._get_generation_config = lambda *a, **k: GenerationConfig()
    hf, _, _ = conv.convert_openai_whisper_to_tfms(str(out / "full.pt"), str(out / "hf"))
    hf = hf.float().eval()
    mel, y_in, y_out = arch_inputs(CKPT_DIMS, 11)
    with torch.no_grad():
        logits = hf(input_features=mel, decoder_input_ids=y_in).logits
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), y_out, label_smoothing=0.1)
    res = {"hf_logits_s17": logits[:, :, ::17].numpy(), "hf_loss": np.float64(loss.item()),  # (every 17th vocabulary column: 0.3 MB)
           "hf_argmax": logits.argmax(-1).numpy()}

    # merge flow on a skeleton of the checkpoint's Linear layers (whisper.load_model is not available; only Linear layers carry
    # adapters, so the skeleton holds exactly those, under the checkpoint's module names)
    from whisper.model import Linear as WLinear
    from whisper_finetune.model.lora import apply_lora, is_lora_enabled, merge_lora

    ckpt = torch.load(out / "lora.pt", map_location="cpu", weights_only=True)
    sd = {k: v.float() for k, v in ckpt["model_state_dict"].items()}
    lin = sorted(k[: -len(".parametrizations.weight.original")] for k in sd if k.endswith(".parametrizations.weight.original"))
    root = torch.nn.Module()
    for name in lin:
        w = sd[name + ".parametrizations.weight.original"]
        mod = root
        parts = name.split(".")
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, torch.nn.Module())
            mod = mod._modules[part]
        mod.add_module(parts[-1], WLinear(w.shape[1], w.shape[0], bias=(name + ".bias") in sd))
    apply_lora(root, lora_config={"rank": 8, "lora_alpha": 16, "lora_dropout": 0.1})
    sub = {k: v for k, v in sd.items() if any(k.startswith(n + ".") for n in lin)}
    missing, unexpected = root.load_state_dict(sub, strict=False)
    assert not missing and not unexpected, (missing, unexpected)  # the reference raises on either (merge_lora_weights.py:45-50)
    assert is_lora_enabled(root)
    root.eval()
    merge_lora(root)
    assert not is_lora_enabled(root)
    merged = dict(root.state_dict())
    for name in lin:
        res["merged::" + name] = merged[name + ".weight"].numpy()
    res["linear_names"] = np.array(lin)
    res["lora_keys"] = np.array(sorted(k for k in sd if "lora" in k))
    np.savez_compressed(HERE / "ref_checkpoint.npz", **res)
    print("ref_checkpoint.npz: hf loss", loss.item(), "merged Linears", len(lin))


def gen_ref_checkpoint():
    import subprocess

    d = tempfile.mkdtemp()
    subprocess.run([sys.executable, __file__, "gen_ref_checkpoint_make", d], check=True)
    subprocess.run([sys.executable, __file__, "gen_ref_checkpoint_consume", d], check=True)


if __name__ == "__main__":
    import subprocess

    if len(sys.argv) > 1:  # one generator per process: several of them import the reference under different stubs
        globals()[sys.argv[1]]()
        sys.exit(0)
    for fn in ("gen_ref_sched", "gen_ref_train_step", "gen_ref_dataset", "gen_ref_checkpoint"):
        subprocess.run([sys.executable, __file__, fn], check=True)
    gen_ref_optim()
    gen_ref_eval()
    gen_arch()
    gen_logmel()
    gen_ref_host()
