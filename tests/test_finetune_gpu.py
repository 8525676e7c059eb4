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
