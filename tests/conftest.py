import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT / "whisper-finetune_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_arch():
    return np.load(GOLDEN / "whisper_arch.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def golden_logmel():
    return np.load(GOLDEN / "logmel.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def golden_host():
    return np.load(GOLDEN / "ref_host.npz", allow_pickle=False)
