"""Developer scripts that use WFT_GEMM_DIAG or the A/B variables import this first: it points WFT_LIB at libwft_timing.so
(`make -C whisper-finetune_amd/csrc TIMING=1`, -DWFT_TIMING_BUILDS) — the shipped libwft.so does not read those variables."""
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parents[2] / "whisper-finetune_amd"
_LIB = _PKG / "libwft_timing.so"
if "WFT_LIB" not in os.environ:
    if not _LIB.exists():
        subprocess.run(["make", "-C", str(_PKG / "csrc"), "TIMING=1", "-j", "6"], check=True)
    os.environ["WFT_LIB"] = str(_LIB)
