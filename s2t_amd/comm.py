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
    # The ranks should leave this function the same way.  What is exchanged over the host group (object collectives: they
    # work on any backend, gloo or nccl) BEFORE anyone enters the blocking s2t_comm_init: whether each rank can load the
    # library at all, and rank 0's (status, id).  What cannot be covered: a rank that dies or hangs INSIDE s2t_comm_init
    # (ncclCommInitRank) leaves the others blocked there — RCCL's own rendezvous, not ours.  After the call the ranks agree
    # on one verdict, and a rank whose communicator came up while another's did not destroys its own.
    buf = C.create_string_buffer(128)
    try:
        lib = L.lib()
        load_err = None
    except Exception as e:  # noqa: BLE001 — reported to every rank below
        lib, load_err = None, "%s: %s" % (type(e).__name__, e)
    rc0 = lib.s2t_comm_unique_id(buf) if (rank == 0 and lib is not None) else 0
    if world > 1:
        states = [None] * world
        dist.all_gather_object(states, (load_err, int(rc0), bytes(buf.raw) if rank == 0 else None))
        bad = [(r, st[0]) for r, st in enumerate(states) if st[0] is not None]
        if bad:
            raise RuntimeError("s2t_amd.comm: libs2t_hip is not loadable on rank(s) %s" % bad)
        rc0, raw = states[0][1], states[0][2]
        buf = C.create_string_buffer(raw, 128)
    elif load_err is not None:
        raise RuntimeError(load_err)
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
        try:
            verdicts = [None] * world
            dist.all_gather_object(verdicts, int(rc))  # (an object collective: any backend of the default group carries it)
        except Exception:
            if rc == 0:
                L.lib().s2t_comm_destroy()  # the exchange itself failed: do not leak a communicator nobody will use
            raise
        if any(v != 0 for v in verdicts) and rc == 0:
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
