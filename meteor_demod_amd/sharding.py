"""Multi-GPU layer: one process per GPU, streams sharded across ranks.

The path shards only ACROSS streams (recordings, or tiles treated as streams):
within a stream it is a serial recurrence.  So there is no
collective on the data path at all — each rank demodulates its own contiguous
range of streams from its own HBM.  The only exchange the north star names is the
optional fan-in of the soft-symbol buffers to one rank, done here with
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests): one all_gather of per-stream symbol counts, then point-to-point
send/recv of exactly the int8 rows each rank holds, at the nominal symbol pitch
(compacted on the device first).
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

    One all_gather of (streams, pitch) per rank, one all_gather of the symbol counts, then point-to-point transfers of EXACTLY
    the rows each rank holds ("ncclSend/ncclRecv, variable sizes"): nothing is padded to the largest shard or the
    longest pitch on the wire (round 6; rounds 1-5 used one padded gather), and a rank whose pitch is the common one lands
    directly in its slice of the result (no staging copy on the root).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = soft_local.device
    n_local, pitch_local = int(soft_local.shape[0]), int(soft_local.shape[1])
    lo, hi = shard_range(n_streams, rank, world)
    if n_local != hi - lo:
        raise ValueError(f"rank {rank} holds {n_local} streams, its shard of {n_streams} over {world} ranks is {hi - lo}")
    n_max = max(shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0] for r in range(world))

    # 1) who holds what: (streams, pitch) of every rank, and the counts (padded to the largest shard: 4 B per stream)
    shape = torch.tensor([n_local, pitch_local], dtype=torch.int64, device=dev)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    shapes = [(int(t[0]), int(t[1])) for t in torch.stack(shapes).cpu()]
    cap = max(p for _, p in shapes)
    cnt_pad = torch.zeros(n_max, dtype=torch.int32, device=dev)
    cnt_pad[:n_local] = counts_local.to(torch.int32)
    all_cnt = [torch.empty_like(cnt_pad) for _ in range(world)]
    dist.all_gather(all_cnt, cnt_pad, group=group)

    # 2) the rows, point to point, exact sizes
    if rank != dst:
        if n_local:
            send = soft_local if soft_local.is_contiguous() else soft_local.contiguous()
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, _global_rank(dst, group), group)]):
                w.wait()
        return None, None

    soft = torch.empty((n_streams, cap, 2), dtype=torch.int8, device=dev)
    ops, staged = [], []
    for r in range(world):
        rlo, rhi = shard_range(n_streams, r, world)
        n_r, pitch_r = shapes[r]
        if n_r != rhi - rlo:
            raise ValueError(f"rank {r} announced {n_r} streams, its shard is {rhi - rlo}")
        if n_r == 0:
            continue
        if r == rank:
            soft[rlo:rhi, :pitch_r] = soft_local
            if pitch_r < cap:
                soft[rlo:rhi, pitch_r:] = 0
        elif pitch_r == cap:
            ops.append(dist.P2POp(dist.irecv, soft[rlo:rhi], _global_rank(r, group), group))      # a contiguous slice: lands in place
        else:
            tmp = torch.empty((n_r, pitch_r, 2), dtype=torch.int8, device=dev)
            staged.append((rlo, rhi, pitch_r, tmp))
            ops.append(dist.P2POp(dist.irecv, tmp, _global_rank(r, group), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for rlo, rhi, pitch_r, tmp in staged:
        soft[rlo:rhi, :pitch_r] = tmp
        soft[rlo:rhi, pitch_r:] = 0
    counts = torch.cat([all_cnt[r][: shapes[r][0]] for r in range(world)])
    return soft, counts


def _global_rank(group_rank: int, group) -> int:
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def fanin_bytes_on_the_wire(shapes, dst: int = 0) -> int:
    """Bytes the fan-in moves for per-rank (streams, pitch) shapes, the root's own rows excluded: what `fanin_soft` sends, exactly."""
    return sum(2 * n * p for r, (n, p) in enumerate(shapes) if r != dst)


def init_from_env(backend: Optional[str] = None, timeout_s: float = 600.0):
    """torch.distributed init for `python -m torch.distributed.run` launches (env://).  `timeout_s` bounds every collective
    (gloo raises after it; RCCL's watchdog aborts the collective): nothing on this path may wait forever for a rank that died."""
    import os
    from datetime import timedelta
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
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timedelta(seconds=timeout_s))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s))
    return rank, local, world
