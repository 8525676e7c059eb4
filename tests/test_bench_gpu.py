"""bench.py contract on a GPU box: one JSON object as the LAST stdout line with the required keys, plain and through
the DDP wrapper (1-rank RCCL group), on a small model so that it runs in seconds."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.parametrize("force_ddp", [False, True])
def test_bench_prints_one_json_line_last(force_ddp):
    env = dict(os.environ, WFT_BENCH_FORCE_DDP="1" if force_ddp else "0", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--model", "base", "--batch", "4", "--seq", "32", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    out = json.loads(lines[-1])  # the JSON object is the last line even when RCCL prints its banners
    assert REQUIRED <= set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["unit"] == "audio-s/s" and out["dtype"] == "bf16" and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert sum(1 for ln in lines if ln.startswith("{")) == 1


def test_bench_through_the_drivers_launcher_line():
    """The exact command the driver scales (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`) at N = 1, with the DDP wrapper forced on: rendezvous from the launcher's
    environment, a 1-rank RCCL group, the bucketed reducer around the engine, per-tile GEMM launches — the N > 1 code path on
    the one GPU this box has."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WFT_BENCH_FORCE_DDP="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--model", "base", "--batch", "4", "--seq", "32",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[-1])
    assert REQUIRED <= set(out) and out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["parallelism"] == "dp1"
