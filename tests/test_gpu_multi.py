"""The N > 1 path on the ONE GPU a test box has (no process had ever run bench.py's world > 1 branch).

`bench.py --gpus 2 --oversubscribe` starts two ranks through torch.distributed.run exactly as the driver's launch does, puts both
on device 0 and uses gloo where the real run uses RCCL: the spawn (before anything touches the GPU), the rendezvous on 127.0.0.1,
the barrier + max-over-ranks timing, the per-rank kernel time, the fan-in (compact on the device, count all_gather, row gather),
the per-rank check against the oracle and the clean exit of the process tree all execute.  What stays unproven is the RCCL
transport itself (tests/test_gpu_parity.py::test_rccl_fanin_path_two_ranks runs where two GPUs exist)."""
from __future__ import annotations

import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _bench(*args, timeout=900):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, cwd=str(ROOT), timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("config", ["c1", "c3"])
def test_bench_two_ranks_oversubscribed_on_one_gpu(gpu_device, config):
    T, L = 4096, 4160
    r, lines = _bench("--gpus", "2", "--oversubscribe", "--config", config, "--tiles", str(T), "--tile-samples", str(L), "--steps", "2", "--warmup", "1")       # (no --fanin: the gather on rank 0 is the default at N > 1 since round 5)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                                  # rank 0 prints ONE line, the other rank nothing
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["samples_per_step"] == 2 * T * L and d["config"]["tiles_per_gpu"] == T
    assert d["value"] > 0 and abs(d["value"] - 2 * T * L * 2 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.01 * d["value"]
    assert "every rank" in d["check"] and "byte-identical" in d["check"], d["check"]
    per_rank = d["roofline"]["kernel_ms_over_ranks"]
    assert 0 < per_rank["min"] <= per_rank["max"] == d["roofline"]["kernel_ms"]
    assert d["fanin"]["rows_intact"] is True and d["fanin"]["bytes_over_xgmi"] > 0 and d["fanin"]["ms"] > 0 and d["fanin"]["gbytes_per_s"] > 0
    assert "DRY RUN" in d["oversubscribed"]


def test_bench_two_ranks_no_fanin_flag(gpu_device):
    r, lines = _bench("--gpus", "2", "--oversubscribe", "--tiles", "4096", "--tile-samples", "4160", "--steps", "1", "--warmup", "0", "--no-fanin", "--no-check")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[0])
    assert "fanin" not in d and d["n_gpus"] == 2


def test_bench_refuses_more_gpus_than_the_node_has(gpu_device):
    """Without --oversubscribe a request for more GPUs than exist must not print a line labelled with the requested count."""
    import torch
    have = torch.cuda.device_count()
    r, lines = _bench("--gpus", str(have + 1), "--tiles", "4096", "--tile-samples", "4160", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and not lines and "refusing" in r.stderr
