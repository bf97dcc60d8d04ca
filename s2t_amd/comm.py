"""Process-global RCCL communicator behind the C-ABI (include/s2t_hip.h: s2t_comm_*).

One process per GPU.  Rank 0 draws the 128-byte RCCL unique id and hands it to the other ranks over whatever host
channel ``torch.distributed`` already has (its default process group — gloo is enough: no second RCCL communicator
and no NCCL watchdog thread exist then, which is what makes the gradient all-reduce capturable in a hipGraph).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L

_STATE = {"world": 0}


def initialized() -> bool:
    return _STATE["world"] > 0


def world_size() -> int:
    return _STATE["world"]


def init(rank=None, world=None, device=None):
    """Create the communicator (collective: every rank calls it).  Uses torch.distributed only to pass the id around;
    ``world == 1`` needs no torch.distributed at all."""
    if initialized():
        return
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if device is not None:
        torch.cuda.set_device(device)
    # Every rank must leave this function the same way: a failure on ONE rank (no loadable librccl on rank 0, a communicator
    # that does not come up on rank 3) may not leave the others blocked in a collective.  Rank 0 therefore broadcasts
    # (status, id) — not the bare id — and after s2t_comm_init the ranks agree on a common verdict over the host group.
    buf = C.create_string_buffer(128)
    rc0 = L.lib().s2t_comm_unique_id(buf) if rank == 0 else 0
    if world > 1:
        box = [(int(rc0), bytes(buf.raw)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rc0, raw = box[0]
        buf = C.create_string_buffer(raw, 128)
    L.check(rc0, "s2t_comm_unique_id (on rank 0)")
    # RCCL prints a version banner to stdout when a communicator is created; stdout belongs to the caller (bench.py
    # prints ONE JSON line there), so fd 1 points at stderr for the duration of the call
    import os
    import sys

    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        rc = L.lib().s2t_comm_init(int(rank), int(world), buf)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    if world > 1:  # the worst status of any rank becomes everyone's: all take the library communicator or none does
        verdict = torch.tensor([abs(int(rc))], dtype=torch.int64)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX)
        if int(verdict) != 0 and rc == 0:
            L.lib().s2t_comm_destroy()
            rc = -4  # S2T_ERR_UNSUPPORTED: another rank has no communicator
    L.check(rc, "s2t_comm_init (some rank)")
    _STATE["world"] = world


def all_reduce_(t: torch.Tensor, average=True):
    """In-place all-reduce of a contiguous CUDA tensor on the CURRENT stream."""
    assert t.is_cuda and t.is_contiguous()
    L.check(L.lib().s2t_allreduce_bucket(t.data_ptr(), t.numel(), L.dtype_id(t.dtype), int(bool(average)), L.stream_ptr()),
            "s2t_allreduce_bucket")
    return t


def destroy():
    if initialized():
        L.check(L.lib().s2t_comm_destroy(), "s2t_comm_destroy")
        _STATE["world"] = 0
