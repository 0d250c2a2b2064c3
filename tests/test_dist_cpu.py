"""gloo tests (world size 2, 4 and 8: what the driver's scaling run uses) of the multi-GPU layer (sharding + soft-symbol fan-in).

The demodulation itself needs a GPU, so each rank uses the oracle as a stand-in
producer of per-stream soft symbols; what is tested is that sharding N streams
over ranks and fanning the results in gives exactly the single-process result.
"""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.sharding import fanin_soft, owner_of, shard_range

N_SAMPLES = 6000


def _stream_soft(idx: int):
    import oracle_py as O
    cfg = DemodConfig(samplerate=230000)
    st = synth.make_stream(900 + idx, 230000, 72000, f0_hz=100.0 * idx, esn0_db=20.0)
    iq = synth.generate_host(st, N_SAMPLES - 350 * idx)          # ragged lengths (750 samples for stream 15)
    return O.oracle_demod(cfg, iq)[0]


def _worker(rank: int, world: int, port: int, q, N_STREAMS: int):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(N_STREAMS, rank, world)
    softs = [_stream_soft(i) for i in range(lo, hi)]
    cap = max([s.shape[0] for s in softs] + [1])
    local = torch.zeros((hi - lo, cap, 2), dtype=torch.int8)
    counts = torch.zeros(hi - lo, dtype=torch.int32)
    for k, s in enumerate(softs):
        local[k, : s.shape[0]] = torch.from_numpy(s)
        counts[k] = s.shape[0]
    soft, cnt = fanin_soft(local, counts, N_STREAMS, dst=0)
    if rank == 0:
        q.put((soft.numpy(), cnt.numpy()))
    else:
        assert soft is None and cnt is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_partition_the_streams():
    for n in (0, 1, 5, 8, 64, 65537):
        for w in (1, 2, 3, 8):
            ranges = [shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
            for s in range(0, n, max(1, n // 17)):
                r = owner_of(s, n, w)
                assert ranges[r][0] <= s < ranges[r][1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,N_STREAMS", [(2, 5), (2, 6), (4, 11), (4, 3), (8, 16), (8, 13)],
                         ids=["2-uneven-shards", "2-even-shards-zero-copy", "4-uneven", "4-a-rank-without-streams", "8-even-zero-copy", "8-uneven"])
def test_fanin_equals_single_process(world, N_STREAMS):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, N_STREAMS)) for r in range(world)]
    for p in procs:
        p.start()
    soft, cnt = q.get(timeout=150)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i in range(N_STREAMS):
        want = _stream_soft(i)
        assert cnt[i] == want.shape[0]
        assert np.array_equal(soft[i, : cnt[i]], want)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` starts its own ranks; with fewer than N devices it must fail loudly instead of printing an
    N=1 line (the driver's scaling run degenerated to one GPU)."""
    import subprocess
    import sys
    from pathlib import Path
    if torch.cuda.device_count() >= 64:
        pytest.skip("needs a node with fewer than 64 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--gpus", "64", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
