"""Data-parallel gradient all-reduce over RCCL/xGMI for the flat gradient buffer.

Mirrors the protocol of the reference's wrapper (fairseq/distributed/legacy_distributed_data_parallel.py:29-160,
hooked from trainer.py:714-718 through optim/fairseq_optimizer.py:97-100): ``forward`` delegates to the wrapped
module, ``no_sync()`` suppresses the reduction while accumulating, ``all_reduce_grads()`` leaves every gradient equal
to ``sum_over_ranks(grad) / world_size`` (the trainer then multiplies by ``world / sample_size``, trainer.py:729-734).

What is different (MI355X-first):
  * gradients already live in ONE contiguous fp32 buffer (flat_params.py), so a bucket is a *view* — the reference's
    copy-in / div_ / all_reduce / copy-out (:82-120) becomes a single in-place collective per bucket, with the
    division folded into ``ReduceOp.AVG`` on RCCL;
  * buckets are reduced while backward is still running: the autograd functions report each parameter whose
    gradient is final (functional._ready); when the last parameter of a bucket has reported, its all-reduce is
    launched on a side HIP stream behind an event.  xGMI is point-to-point (7 links per GPU): a few large buckets
    (default 32 MiB) keep every link streaming instead of many latency-bound small rings;
  * BatchNorm buffers are NOT synchronised, like the reference (":29-31 does not broadcast buffers");
  * on the GPU the collective is the library's own RCCL communicator (s2t_amd/comm.py -> s2t_allreduce_bucket): a plain
    stream-ordered launch with no watchdog thread, so the whole update — backward, the bucketed all-reduce on its side
    stream, clip and Adam — is ONE captured hipGraph.  ``torch.distributed`` collectives remain for CPU tensors (the
    gloo tests) and when no communicator was created.
"""
from contextlib import contextmanager
from typing import Dict, List

import torch
import torch.distributed as dist
import torch.nn as nn

from . import comm as Comm
from . import functional as Fn


class LegacyDistributedDataParallel(nn.Module):
    def __init__(self, module, process_group=None, buffer_size=2 ** 23, overlap=True, single_rank_collectives=False,
                 reduce_dtype=None):
        """``buffer_size``: bucket size in ELEMENTS (2**23 fp32 = 32 MiB).
        ``single_rank_collectives``: issue the (trivial) collectives even when the group has one rank, so that the
        RCCL + side-stream + hipGraph-capture path can be exercised on a single GPU.
        ``reduce_dtype`` = torch.bfloat16: the buckets travel in bf16 (half the bytes on the xGMI links) — what the reference
        does under ``--fp16``, where the wrapper's buffer has the parameters' half-precision dtype
        (legacy_distributed_data_parallel.py:44-48,82-120): fp32 bucket -> bf16 staging copy -> all-reduce (sum) -> back to
        fp32 divided by the world size.  Default (None, or S2T_DDP_REDUCE=bf16 in the environment): fp32."""
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if Comm.initialized():
            self.world_size = Comm.world_size()
        self._idle = self.world_size == 1 and not single_rank_collectives
        self.accumulate_grads = False
        self.overlap = overlap
        flat = module.flat
        assert flat is not None, "call model.prepare() before wrapping it"
        self.flat = flat
        # buckets = contiguous slices of the flat gradient buffer, filled from the END (backward order)
        n = flat.numel
        bounds = list(range(n, 0, -buffer_size))[::-1]
        starts = [max(0, b - buffer_size) for b in bounds]
        self.buckets = [(s, e) for s, e in zip(starts, bounds)]
        self._bucket_of: Dict[int, List[int]] = {}
        for p in flat.params:
            o, cnt = flat.offsets[id(p)], p.numel()
            self._bucket_of[id(p)] = [i for i, (s, e) in enumerate(self.buckets) if o < e and o + cnt > s]
        self._expected: Dict[int, int] = {}  # ready-calls per parameter, learned on the first step
        self._seen: Dict[int, int] = {}
        self._pending = None
        self._launched = set()
        self._work = []
        self._side = torch.cuda.Stream() if (flat.grad.is_cuda and overlap) else None
        self._learning = True
        import os

        if reduce_dtype is None and os.environ.get("S2T_DDP_REDUCE", "") == "bf16":
            reduce_dtype = torch.bfloat16
        assert reduce_dtype in (None, torch.float32, torch.bfloat16)
        self.reduce_dtype = reduce_dtype if reduce_dtype is not None else torch.float32
        # one bf16 staging block per bucket (buckets may be in flight together on the side stream)
        self._stage = [torch.empty((e - s + 3) // 4 * 4, dtype=torch.bfloat16, device=flat.grad.device)
                       for s, e in self.buckets] if self.reduce_dtype == torch.bfloat16 else None
        # RCCL's all-reduce kernels run BESIDE backward and take compute units.  The fused feed-forward kernels deal a row block
        # to several workgroups that wait for each other (csrc/ffn_pc.hip), sized for a grid that is resident at once.  With CUs
        # held by another kernel the late workgroups' partners are dispatched within 64 blocks of them and the grid drains in
        # dispatch order, so the exchange completes (rehearsed with 32 and 64 CUs held: tests/test_rowblock_gpu.py::
        # test_split_ffn_forms_with_compute_units_held_by_another_kernel) — at the price of a second round of workgroups.
        # S2T_COMM_CUS=n lowers the budget the split is sized from (s2t_ffn_cu_budget) by n instead: fewer, longer workgroups
        # in one round.  Which is faster beside a real all-reduce is a measurement for the 8-GPU node; default: leave the split.
        if flat.grad.is_cuda and self.world_size > 1 and overlap and os.environ.get("S2T_COMM_CUS"):
            from . import kernels as K
            K.ffn_cu_budget(max(64, K.ffn_cu_budget(0) - int(os.environ["S2T_COMM_CUS"])))

    # -- module protocol ---------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)  # distributed/module_proxy_wrapper.py behaviour

    @contextmanager
    def no_sync(self):
        old = self.accumulate_grads
        self.accumulate_grads = True
        try:
            yield
        finally:
            self.accumulate_grads = old

    # -- overlap machinery ---------------------------------------------------------------------------
    @property
    def active(self):
        """True when all_reduce_grads() issues collectives."""
        return not self._idle

    def begin_backward(self, overlap=True):
        """Arm the grad-ready hook for one backward pass (``overlap=False``: reduce everything in all_reduce_grads)."""
        self._seen = {}
        self._launched = set()
        self._work = []
        if self._idle or self.accumulate_grads:
            Fn._HOOKS["grad_ready"] = None
            return
        if self._learning:
            Fn._HOOKS["grad_ready"] = self._count
        elif self.overlap and overlap:
            self._pending = [0] * len(self.buckets)
            for p in self.flat.params:
                for b in self._bucket_of[id(p)]:
                    self._pending[b] += self._expected.get(id(p), 0)
            Fn._HOOKS["grad_ready"] = self._on_ready
        else:
            Fn._HOOKS["grad_ready"] = None

    def _count(self, p):
        self._seen[id(p)] = self._seen.get(id(p), 0) + 1

    def _on_ready(self, p):
        for b in self._bucket_of[id(p)]:
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        view = self.flat.grad[s:e]
        self._launched.add(b)
        wg = Fn.wgrad_stream() if view.is_cuda else None  # weight gradients are produced on their own stream
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            if wg is not None:
                self._side.wait_stream(wg)
            with torch.cuda.stream(self._side):
                self._reduce(view, b)
        else:
            if wg is not None:
                torch.cuda.current_stream().wait_stream(wg)
            self._reduce(view, b)

    def _reduce(self, view, b=None):
        if self._stage is not None and b is not None:
            # bf16 on the wire: cast, SUM all-reduce, cast back with the 1 / world division
            st = self._stage[b][:view.numel()] if view.numel() % 4 == 0 else self._stage[b]
            n = view.numel()
            if view.is_cuda and n % 4 == 0:
                from . import kernels as K

                K.cast_f32_to_bf16(view, st, n)
                if Comm.initialized():
                    Comm.all_reduce_(st, average=False)
                else:
                    dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.process_group)
                K.cast_bf16_to_f32(st, view, n, 1.0 / self.world_size)
            else:
                st = st[:n]
                st.copy_(view)
                dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.process_group)
                view.copy_(st.float() / self.world_size)
            return
        if view.is_cuda and Comm.initialized():
            Comm.all_reduce_(view, average=True)
        elif view.is_cuda and dist.get_backend(self.process_group) == "nccl":
            dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.process_group)
        else:  # gloo (CPU tests, multi-rank rehearsals on one GPU): no AVG
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.process_group)
            view.div_(self.world_size)

    def all_reduce_grads(self):
        """Finish the reduction: every gradient becomes sum_over_ranks / world_size (reference :76-160)."""
        Fn._HOOKS["grad_ready"] = None
        if self._idle or self.accumulate_grads:
            return
        if self._learning:
            self._expected = dict(self._seen)
            self._learning = False
        for b in range(len(self.buckets) - 1, -1, -1):
            if b not in self._launched:
                self._launch(b)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
