"""Multi-GPU layer: one process per GPU, streams sharded across ranks.

The path shards only ACROSS streams (recordings, or tiles treated as streams):
within a stream it is a serial recurrence.  So there is no
collective on the data path at all — each rank demodulates its own contiguous
range of streams from its own HBM.  The only exchange the north star names is the
optional fan-in of the soft-symbol buffers to one rank, done here with
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests): one all_gather of per-stream symbol counts, then one gather of
the int8 rows at the nominal symbol pitch (compacted on the device first).
"""
from __future__ import annotations

from typing import Optional


def shard_range(n_streams: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of the streams owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(stream: int, n_streams: int, world: int) -> int:
    base, extra = divmod(n_streams, world)
    edge = extra * (base + 1)
    if stream < edge:
        return stream // (base + 1)
    return extra + (stream - edge) // base if base else world - 1


def fanin_soft(soft_local, counts_local, n_streams: int, dst: int = 0, group=None):
    """Collect every rank's soft symbols on `dst`.

    soft_local  : int8 tensor [n_local, pitch, 2] (device tensor for nccl, CPU for gloo).  Pass rows at the NOMINAL pitch
                  (Demodulator.compact / mdemod_compact_soft: 0.63 B per input sample at 72k in 230 kS/s), not at the
                  hard-bound capacity the kernels write with (one symbol per input sample: 3.2x the bytes).
    counts_local: int32 tensor [n_local], symbols valid per local stream
    Returns on dst: (soft [n_streams, pitch_max, 2], counts [n_streams]); elsewhere (None, None).

    One all_reduce (common pitch), one all_gather (counts), one gather (symbols).  The gather lands directly in the
    result buffer when the streams divide evenly over the ranks (no staging copy on the root).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = soft_local.device
    n_local = soft_local.shape[0]
    n_max = max(shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0] for r in range(world))

    # 1) agree on a common row pitch (ranks may have been given different block lengths)
    cap = torch.tensor([soft_local.shape[1]], dtype=torch.int64, device=dev)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
    cap = int(cap.item())

    cnt_pad = torch.zeros(n_max, dtype=torch.int32, device=dev)
    cnt_pad[:n_local] = counts_local.to(torch.int32)
    all_cnt = [torch.empty_like(cnt_pad) for _ in range(world)]
    dist.all_gather(all_cnt, cnt_pad, group=group)

    # 2) rows: padded only if this rank holds fewer streams or a shorter pitch than the largest shard
    if n_local == n_max and soft_local.shape[1] == cap and soft_local.is_contiguous():
        send = soft_local
    else:
        send = torch.zeros((n_max, cap, 2), dtype=torch.int8, device=dev)
        send[:n_local, : soft_local.shape[1]] = soft_local
    even = n_streams == n_max * world
    out = torch.empty((n_max * world, cap, 2), dtype=torch.int8, device=dev) if rank == dst else None
    gathered = [out[r * n_max:(r + 1) * n_max] for r in range(world)] if rank == dst else None
    dist.gather(send, gathered, dst=dst, group=group)
    if rank != dst:
        return None, None

    if even:
        return out, torch.cat(all_cnt)
    soft = torch.empty((n_streams, cap, 2), dtype=torch.int8, device=dev)
    counts = torch.empty(n_streams, dtype=torch.int32, device=dev)
    for r in range(world):
        lo, hi = shard_range(n_streams, r, world)
        soft[lo:hi] = gathered[r][: hi - lo]
        counts[lo:hi] = all_cnt[r][: hi - lo]
    return soft, counts


def init_from_env(backend: Optional[str] = None):
    """torch.distributed init for `python -m torch.distributed.run` launches (env://)."""
    import os
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
