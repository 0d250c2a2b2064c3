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


def test_fanin_peer_gathers_the_rows_of_several_contexts(gpu_device):
    """mdemod_fanin_peer - the fan-in of a host that drives several GPUs from one process (the C host's --devices layout): every
    context's GPU writes its compacted rows and counts straight into ONE buffer.  Two contexts of uneven size; both on this box's
    one GPU, the second on device 1 where there is one (then its stores cross xGMI).  The gathered rows are the oracle's bytes, the
    counts the contexts' own, nothing lands outside the rows a context was given."""
    import numpy as np
    import torch
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_py as O
    from meteor_demod_amd import DemodConfig, Demodulator, synth, _capi
    cfg = DemodConfig(samplerate=230000)
    n, sizes = 9000, (5, 3)
    devs = (0, 1 if torch.cuda.device_count() >= 2 else 0)
    streams = [synth.make_stream(60 + i, 230000, 72000, f0_hz=70.0 * i, esn0_db=18.0) for i in range(sum(sizes))]
    ctxs = [Demodulator(cfg, sizes[k], device=devs[k]) for k in range(2)]
    try:
        pitch = ctxs[0].nominal_pitch(n)
        rows = sum(sizes)
        CAN = 0x5A
        big = torch.full((rows + 2, pitch, 2), CAN, dtype=torch.int8, device="cuda:0")     # one canary row either side
        counts = torch.full((rows + 2,), 0x7FFFFFFF, dtype=torch.int32, device="cuda:0")
        at = 0
        for k, d in enumerate(ctxs):
            with torch.cuda.device(devs[k]):
                x = synth.generate_device(streams[at: at + sizes[k]], n, device=devs[k])
                soft = d.process(x)
                d.fanin_peer(soft, pitch, big, 1 + at, counts)
                torch.cuda.synchronize(devs[k])
            at += sizes[k]
        got, cnt = big.cpu().numpy(), counts.cpu().numpy()
        assert (got[0] == CAN).all() and (got[-1] == CAN).all() and cnt[0] == 0x7FFFFFFF and cnt[-1] == 0x7FFFFFFF
        for i, st in enumerate(streams):
            want = O.oracle_demod(cfg, synth.generate_host(st, n))[0]
            assert cnt[1 + i] == want.shape[0] and np.array_equal(got[1 + i, : want.shape[0]], want), i
        # refusals: a pitch that is not a multiple of 8, rows that do not fit, a device that is not there
        with pytest.raises(ValueError):
            ctxs[0].fanin_peer(soft if devs[1] == 0 else ctxs[0].process(synth.generate_device(streams[:5], n)), pitch, big, rows, counts)
        lib = _capi.lib()
        s0 = ctxs[0].process(synth.generate_device(streams[:5], n, device=0))
        assert lib.mdemod_fanin_peer(ctxs[0]._ctx, s0.data_ptr(), s0.shape[1], 0, big.data_ptr(), pitch + 4, 1, None, None) == _capi.MDEMOD_ERR_PARAM
        assert lib.mdemod_fanin_peer(ctxs[0]._ctx, s0.data_ptr(), s0.shape[1], torch.cuda.device_count() + 3, big.data_ptr(), pitch, 1, None, None) == _capi.MDEMOD_ERR_HIP
        assert _capi.last_error() != ""
    finally:
        for d in ctxs:
            d.close()
