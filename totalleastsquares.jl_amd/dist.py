"""Row-sharded multi-GPU glue: one process per GPU (torch.distributed launch), rows split in contiguous
blocks, the library's own RCCL communicator for the N x N Gram exchange.

torch.distributed is used only for rendezvous (broadcasting the RCCL unique id) and for barriers; the
data path collective (ncclAllReduce of the Gram matrix on the handle's stream) lives in libtlsqhip.so.
Works with the `gloo` backend on CPU for everything except the device calls themselves.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def row_partition(M: int, world: int, rank: int):
    """contiguous row block [lo, hi) of rank `rank`; sizes differ by at most one row"""
    base, rem = divmod(M, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def exchange_unique_id(make_id, rank: int, world: int):
    """rank 0 creates the 128-byte RCCL unique id (make_id()), everyone receives it through torch.distributed"""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    uid = box[0]
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def init_engine_comm(engine, rank: int, world: int, force: bool = False):
    """join `engine` (one GPU) to the row-shard communicator.  force: create a one-rank RCCL communicator as well (the switch
    FORCE_COMM has to be set: a single rank normally needs none) - how a one-GPU box exercises the path of an N-GPU run"""
    if world <= 1 and not force:
        return
    uid = exchange_unique_id(engine.unique_id, rank, world)
    engine.comm_init(world, rank, uid)
