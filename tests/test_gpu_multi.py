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


def _bench(*args, timeout=900, env=None):
    import os
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, cwd=str(ROOT), timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def _line_ok(line, n1):
    sys.path.insert(0, str(ROOT / "tests"))
    from test_host_logic import _assert_line_ok, _bench_module
    return _assert_line_ok(line, _bench_module(), n1=n1)


@pytest.mark.parametrize("config", ["c1", "c3"])
def test_bench_two_ranks_oversubscribed_on_one_gpu(gpu_device, config):
    T, L = 4096, 4160
    r, lines = _bench("--gpus", "2", "--oversubscribe", "--config", config, "--tiles", str(T), "--tile-samples", str(L), "--steps", "2", "--warmup", "1")       # (no --fanin: the gather on rank 0 is the default at N > 1 since round 5)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                                  # rank 0 prints ONE line, the other rank nothing
    assert r.stdout.strip().splitlines()[-1] == lines[0]              # ... and it is the LAST thing on stdout
    d = _line_ok(lines[0], n1=False)
    assert "errors" not in d, d["errors"]
    assert d["rccl"]["ranks_seen"] == 2 and d["rccl"]["world"] == 2 and len(d["rccl"]["devices"]) == 2 and d["rccl"]["backend"] == "gloo"
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


@pytest.mark.parametrize("fault,deadline", [("fanin_raise", "120"), ("check_raise", "120"), ("fanin_hang@1", "25"), ("rccl_raise@0", "25")])
def test_bench_line_survives_a_fault_after_the_timed_region(gpu_device, fault, deadline):
    """whatever fails after the timed region at N > 1 - an exception on every rank, an exception on one rank
    (the others then wait in a collective nobody completes), a rank that hangs - the line still appears, once, rc 0, with the
    timed fields and the error text."""
    T, L = 4096, 4160
    r, lines = _bench("--gpus", "2", "--oversubscribe", "--tiles", str(T), "--tile-samples", str(L), "--steps", "2", "--warmup", "1",
                      env={"MDEMOD_BENCH_FAULT": fault, "MDEMOD_BENCH_POST_DEADLINE_S": deadline}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = _line_ok(lines[0], n1=False)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert d["config"]["samples_per_step"] == 2 * T * L
    stage = fault.split("_")[0]
    assert stage in d["errors"] or "post_region" in d["errors"], d["errors"]
    if fault == "fanin_raise":
        assert "injected fault" in d["fanin"]["error"] and "byte-identical" in d["check"]      # the later stages still ran


def test_bench_default_line_is_compact(gpu_device):
    """The N = 1 line at a small shape with the CPU leg on: required keys first, roofline and cpu_baseline complete, under the
    hard bound, the long record in bench_extras.json."""
    r, lines = _bench("--tiles", "8192", "--tile-samples", "4160", "--steps", "2", "--warmup", "1", "--no-check")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]
    d = _line_ok(lines[0], n1=True)
    full = json.loads((ROOT / "bench_extras.json").read_text())
    assert full["value"] == d["value"] and "other_configs" in full and "host_fed" in full
    assert "bench.py full record: {" in r.stderr
