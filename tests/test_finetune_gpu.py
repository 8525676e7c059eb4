"""End-to-end: the finetune.py entrypoint under the reference's YAML schema on synthetic data (one process)."""
from pathlib import Path

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_finetune_entrypoint_synthetic(tmp_path):
    import whisper_finetune.runtime as rt
    from whisper_finetune.scripts import finetune
    cfg = yaml.safe_load((ROOT / "configs" / "DEBUG_synthetic.yaml").read_text())
    cfg["save_dir"] = str(tmp_path)
    losses = finetune.main(cfg)
    t = cfg["training"]
    # 16 clips / batch 2 = 8 micro-batches per epoch, x2 epochs / accum 2 -> 8 optimizer steps; eval every 0.5 epoch -> 2
    assert t["train_steps"] == 8 and t["val_steps"] == 2 and t["global_accum_grad_steps"] == 2 and len(losses) == 8
    assert all(torch.isfinite(torch.tensor(l)) for l in losses) and losses[-1] < losses[0]
    run_dir = next(Path(tmp_path).iterdir())
    ck = torch.load(run_dir / "last_model.pt", map_location="cpu")
    assert ck["dims"]["n_audio_state"] == 384 and "decoder.token_embedding.weight" in ck["model_state_dict"]
    rt.cleanup()


def test_gpu_data_path_yields_reference_batch_layout():
    from whisper_finetune.data.data_loader import SimpleTokenizer, SyntheticDataset, get_dataloader
    torch.manual_seed(0)
    loader = get_dataloader(SyntheticDataset(5, with_timestamps=True), SimpleTokenizer(), batch_size=2, n_mels=80, shuffle=False,
                            no_timestamp_training=False, no_timestamps_rate=0.5, prompt_use_rate=0.0, device=torch.device("cuda:0"),
                            spec_augment=True, spec_augment_params={"time_mask_param": 100, "freq_mask_param": 27, "time_warp_w": 80, "p": 1.0})
    mel, y_in, y_out = next(iter(loader))
    assert mel.shape == (2, 80, 3000) and mel.dtype == torch.float32 and mel.is_cuda
    assert y_in.shape == y_out.shape and y_in.dtype == torch.int64
    assert y_in[0, 0] == 50258 and y_in[0, 1] == 50261 and y_in[0, 2] == 50359  # sot, <|de|>, <|transcribe|>
    assert (y_out == 50257).any()  # eot ends every target sequence
    assert (mel == 0).any()        # SpecAugment masks are zero-valued


@pytest.mark.parametrize("variant", ["lora_muon_sd_recompute", "decoder_only_lora", "lora_adamw_8bit"])
def test_finetune_entrypoint_lora_variants(tmp_path, variant):
    """The entrypoint with the reference's other switches on the synthetic provider: LoRA (dropout 0.1) + Muon/AuxAdam +
    gradient-checkpointing flags (-> the CheckpointedStochastic classes, stochastic depth 0.1) + block recompute + deep
    SpecAugment + cosine warm restarts; and a decoder-only LoRA run (frozen encoder: stochastic depth forced to 0 there,
    scripts/finetune.py:413-416)."""
    import whisper_finetune.runtime as rt
    from whisper_finetune.scripts import finetune
    cfg = yaml.safe_load((ROOT / "configs" / "DEBUG_synthetic.yaml").read_text())
    cfg["save_dir"] = str(tmp_path)
    cfg["dataset"]["synthetic"] = {"train": 8, "val": 2}
    cfg["model"].update({"lora": True, "lora_config": {"rank": 8, "lora_alpha": 16, "lora_dropout": 0.1}})
    t = cfg["training"]
    t.update({"stochastic_depth": 0.1, "gradient_checkpointing_encoder": True, "gradient_checkpointing_decoder": True})
    if variant == "lora_muon_sd_recompute":
        t["wft_recompute"] = True
        cfg["optimizer"] = {"type": "adamw", "muon": True, "8bit": False, "muon_ndim_threshold": 2,
                            "muon_params": {"lr": 2e-4, "momentum": 0.95, "weight_decay": 0.01},
                            "params": {"lr": 2e-4, "weight_decay": 0.01, "betas": [0.9, 0.98], "eps": 1e-6, "amsgrad": False}}
        cfg["lr_scheduler"] = {"type": "cosine_with_warmup_restarts", "warmup_steps": 1, "lr_num_cycles": 2, "lr_gamma": 0.8}
    elif variant == "lora_adamw_8bit":
        # the optimizer block of configs/config_turbo_best.yaml:62-72 (`8bit: True`): bnb.optim.AdamW8bit in the reference,
        # WftAdamW8bit here — block-wise 8-bit moments for the adapter matrices of >= 4 096 elements, fp32 for the small ones
        cfg["optimizer"] = {"type": "adamw", "8bit": True, "params": {"lr": 2e-4, "weight_decay": 0.1, "betas": [0.9, 0.98], "eps": 1e-6, "amsgrad": False}}
    else:
        t["train_only_decoder"] = True
    losses = finetune.main(cfg)
    assert len(losses) == 4 and all(torch.isfinite(torch.tensor(l)) for l in losses)
    run_dir = next(Path(tmp_path).iterdir())
    ck = torch.load(run_dir / "last_model.pt", map_location="cpu")
    keys = set(ck["model_state_dict"])
    assert "decoder.blocks.0.attn.query.parametrizations.weight.0.lora_A" in keys
    assert "decoder.blocks.0.attn.query.parametrizations.weight.0.lora_dropout_mask" in keys
    if variant == "decoder_only_lora":
        assert not any("encoder" in k and "lora" in k for k in keys)  # adapters only on the decoder
    assert (run_dir / "last_model.pt").stat().st_size > 0
    rt.cleanup()


def test_loader_prefetch_path_equals_the_sequential_path():
    """GpuMelLoader with DataLoader workers (num_workers > 0): batch i+1 is staged on a side stream while batch i is consumed.
    Same worker seeds -> same batches as the sequential path (prefetch forced off)."""
    from whisper_finetune.data.data_loader import SimpleTokenizer, SyntheticDataset, get_dataloader
    kw = dict(batch_size=2, n_mels=80, shuffle=False, no_timestamp_training=True, prompt_use_rate=0.0, device=torch.device("cuda:0"),
              spec_augment=True, spec_augment_params={"time_mask_param": 100, "freq_mask_param": 27, "time_warp_w": 80, "p": 1.0},
              num_workers=2)

    def collect(prefetch):
        torch.manual_seed(11)  # DataLoader derives the workers' base seed from the default generator
        loader = get_dataloader(SyntheticDataset(6), SimpleTokenizer(), **kw)
        assert loader.prefetch
        loader.prefetch = prefetch
        return [(m.clone(), a.clone(), b.clone()) for m, a, b in loader]

    seq, pre = collect(False), collect(True)
    assert len(seq) == len(pre) == 3
    for (m0, a0, b0), (m1, a1, b1) in zip(seq, pre):
        assert m1.is_cuda and a1.is_cuda and torch.equal(m0, m1) and torch.equal(a0.to(a1.device), a1) and torch.equal(b0.to(b1.device), b1)
