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
    buf = C.create_string_buffer(128)
    if rank == 0:
        L.check(L.lib().s2t_comm_unique_id(buf), "s2t_comm_unique_id")
    if world > 1:
        box = [bytes(buf.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        buf = C.create_string_buffer(box[0], 128)
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
    L.check(rc, "s2t_comm_init")
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
