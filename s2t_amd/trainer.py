"""Minimal training harness reproducing the ordering of the reference's ``Trainer.train_step``
(fairseq/trainer.py:611-759): zero_grad -> forward/backward -> all-reduce gradients -> multiply by
world/sample_size -> clip by global norm -> Adam step, with the inverse-sqrt schedule of the recipes
(optim/lr_scheduler/inverse_square_root_schedule.py:59-85, egs/mustc/asr/conf/base.yaml:4-9).

Everything after backward is three launches over the flat buffers (sum of squares, clip coefficient, Adam) with
no host synchronisation, so a whole step can be captured in a hipGraph (``capture=True``) — the launch-bound
regime the reference's eager PyTorch loop lives in is removed rather than tuned.
"""
import math

import torch
import torch.distributed as dist

from . import functional as Fn
from . import kernels as K


class Trainer:
    def __init__(self, model, criterion, ddp=None, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0,
                 clip_norm=10.0, warmup_updates=10000, warmup_init_lr=1e-7):
        self.model, self.criterion, self.ddp = model, criterion, ddp
        self.flat = model.flat
        model.shadow_managed = True  # s2t_adam_step rewrites the bf16 shadow with every update
        self.lr, self.betas, self.eps, self.wd, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        self.warmup_updates, self.warmup_init_lr = warmup_updates, warmup_init_lr
        dev = self.flat.master.device
        self.exp_avg = torch.zeros_like(self.flat.master)
        self.exp_avg_sq = torch.zeros_like(self.flat.master)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)  # lr, step_size, grad_scale, grad_norm
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # pinned staging rows for (lr, step_size, world/sample_size): the CPU runs ahead of the stream and the copy reads the row when it
        # EXECUTES, so a row may only be rewritten once its copy has run: one event per row, waited for on wrap-around
        self._hyper_host = torch.zeros(16, 3, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(16, 3)
        self._hyper_ev = [None] * 16
        self.num_updates = 0
        from . import comm as Comm
        self.world = Comm.world_size() if Comm.initialized() else (dist.get_world_size() if dist.is_initialized() else 1)
        self._graph = None
        self._static = None

    # ---- schedule -----------------------------------------------------------------------------------
    def lr_at(self, n):
        """inverse_sqrt: linear warmup_init_lr -> lr over warmup_updates, then lr * sqrt(warmup / n)."""
        if n < self.warmup_updates:
            return self.warmup_init_lr + n * (self.lr - self.warmup_init_lr) / self.warmup_updates
        return self.lr * math.sqrt(self.warmup_updates) / math.sqrt(max(n, 1))

    def _push_hyper(self, sample_size_global):
        """Hyper-parameters of the update that brings the count to ``num_updates + 1``.  The reference sets the learning
        rate AFTER each update (trainer.py:802 -> lr_step_update -> inverse_square_root_schedule.py:69-85), so update t
        runs with lr(t - 1) and the very first one with warmup_init_lr; Adam's bias correction uses t (optim/adam.py:199-204)."""
        t = self.num_updates + 1
        lr = self.lr_at(self.num_updates)
        b1, b2 = self.betas
        slot = t % 16
        ev = self._hyper_ev[slot]
        if ev is not None:
            ev.synchronize()  # the copy that read this row 16 updates ago has executed
        row = self._hyper_host[slot]
        row[0] = lr
        row[1] = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        row[2] = self.world / float(sample_size_global)  # multiply_grads(world / sample_size), trainer.py:729-735
        self.hyper[:3].copy_(row, non_blocking=True)
        if self.hyper.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_ev[slot] = ev

    # ---- one update ----------------------------------------------------------------------------------
    def _fwd_bwd(self, sample, overlap=True):
        # dropout masks are a function of (seed, site, element): the device-resident seed advances once per update
        # (trainer.py:1093-1097 seeds every step with seed + num_updates), sites restart at 0
        Fn.DROPOUT.begin_step(self.flat.master.device)
        Fn.DROPOUT.seed.add_(1)
        self.flat.zero_grad()
        if self.ddp is not None:
            self.ddp.begin_backward(overlap=overlap)
        loss, sample_size, log = self.criterion(self.model, sample, sync_logging=False)
        loss.backward()
        return loss.detach(), log

    def _update(self):
        n = self.flat.numel
        self.sumsq.zero_()
        K.sumsq_accum(self.flat.grad, n, self.sumsq)
        K.clip_coef(self.sumsq, self.clip_norm, -1.0, self.hyper)  # normaliser from hyper[2], see _push_hyper
        K.adam_step(self.flat.master, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.flat.shadow, n, self.betas[0],
                    self.betas[1], self.eps, self.wd, self.hyper)

    def _step_body(self, sample):
        out = self._fwd_bwd(sample)
        if self.ddp is not None:
            self.ddp.all_reduce_grads()  # grads = sum_ranks / world
        self._update()
        return out

    def train_step(self, sample, sample_size_global=None):
        """Eager step.  ``sample_size_global``: sum of sample sizes over ranks (defaults to world * local)."""
        self.model.train()
        if sample_size_global is None:
            sample_size_global = self.world * sample["ntokens"]
        self._push_hyper(sample_size_global)
        loss, log = self._step_body(sample)
        self.num_updates += 1
        K.ffn_exchange_poll()  # raises when a fused feed-forward launch of an EARLIER update timed out in its exchange
        log["gnorm"] = self.hyper[3]
        log["lr"] = self.lr_at(self.num_updates - 1)  # the rate this update ran with
        return loss, log

    # ---- graph-captured step (fixed shapes) -------------------------------------------------------------
    def capture(self, sample, sample_size_global=None, warmup=2):
        """Capture one full update for ``sample``'s shapes into a hipGraph; returns nothing, use ``replay``."""
        self.model.train()
        if sample_size_global is None:
            sample_size_global = self.world * sample["ntokens"]
        self._static = sample
        for mod in self.model.modules():  # CTC-guided compression: training passes keep the frame axis at its bound from here
            if hasattr(mod, "compression_layers") and getattr(mod, "compression_bounded", False) is not True:  # on (no host copy)
                mod.compression_bounded = "train"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._push_hyper(sample_size_global)
                self._step_body(sample)
                self.num_updates += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the grouped weight-gradient tables of the captured step: one pinned/device block per launch (a data-parallel step
        # flushes once per gradient stage; a flush is two launches when some problem needs the 128 x 128 kernel)
        Fn.reserve_wgrad_staging(self.flat.master.device, count=16 if self.ddp is not None else 2)
        Fn.pin_buffers_for_graph()
        # the per-batch bookkeeping of THIS batch object (lengths, masks, positions, packed-row geometry: functional.batch_memo)
        # lives outside the graph, which bakes its addresses in: from here on an eager pass over another batch may not replace or
        # release those tensors (it gets entries of its own), load_batch refreshes them in place
        Fn.unpin_batch_memos(Fn.memo_owner(self))
        Fn.pin_batch_memos(list(self._tensors(sample)), Fn.memo_owner(self))
        self._graph = torch.cuda.CUDAGraph()
        self._graph2 = None
        self._ssg = sample_size_global
        self._push_hyper(sample_size_global)
        from . import comm as Comm
        if self.ddp is not None and self.ddp.active and not Comm.initialized():
            # torch.distributed's RCCL collectives stay OUTSIDE the capture (the process-group watchdog thread polls events, which a
            # capturing stream forbids): graph 1 = forward + backward, eager bucketed all-reduce, graph 2 = clip + Adam.
            # The reduction is then not overlapped with backward (it is in the eager path).
            # capture_error_mode="thread_local": the watchdog thread may still be querying the events of the eager
            # all-reduce that preceded the capture; under the default global mode that query is an error that aborts
            # the process (an intermittent SIGABRT, depending on when the watchdog last polled)
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self._graph_out = self._fwd_bwd(sample, overlap=False)
            self.ddp.all_reduce_grads()
            self._graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph2, pool=self._graph.pool(), capture_error_mode="thread_local"):
                self._update()
        else:
            # single GPU, or the library's own communicator: the whole update — backward, the bucketed all-reduce on its
            # side stream (forked and joined inside the capture), clip and Adam — is ONE graph
            with torch.cuda.graph(self._graph):
                self._graph_out = self._step_body(sample)

    @staticmethod
    def _tensors(dct):
        for v in dct.values():
            if isinstance(v, dict):
                yield from Trainer._tensors(v)
            elif torch.is_tensor(v):
                yield v

    def release(self):
        """Drop the captured step: its graphs, the pins on the per-batch bookkeeping it read, the bounded compression form;
        the model goes back to refreshing its bf16 shadow itself (``release_trainer``)."""
        self._graph = self._graph2 = None
        self._graph_out = None
        self._static = None
        Fn.unpin_batch_memos(Fn.memo_owner(self))
        if hasattr(self.model, "release_trainer"):
            self.model.release_trainer()

    def __del__(self):
        try:
            uid = self.__dict__.get("_s2t_uid")
            if uid is not None:
                Fn.unpin_batch_memos(uid)
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass

    def load_batch(self, sample):
        """Copy a batch of the captured shapes into the static one the graph reads (tensors by key, recursively), then
        redo the per-batch bookkeeping that lives outside the graph (functional.batch_memo)."""
        def fill(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    fill(dst[k], v)
                elif torch.is_tensor(v) and torch.is_tensor(dst.get(k)):
                    if dst[k].shape != v.shape:
                        raise ValueError(f"batch field {k}: shape {tuple(v.shape)} differs from the captured {tuple(dst[k].shape)}")
                    dst[k].copy_(v, non_blocking=True)
        fill(self._static, sample)
        # lengths / positions / target matrices derived from the batch, recomputed in place
        Fn.refresh_batch_memos(list(self._tensors(self._static)))

    def replay(self, sample=None, sample_size_global=None):
        """One captured update; ``sample`` (same shapes as the captured batch) is copied into the static batch first."""
        if sample is not None and sample is not self._static:
            self.load_batch(sample)
            self._ssg = self.world * sample["ntokens"]
        if sample_size_global is not None:
            self._ssg = sample_size_global
        self._push_hyper(self._ssg)
        self._graph.replay()
        if self._graph2 is not None:
            self.ddp.begin_backward(overlap=False)
            self.ddp.all_reduce_grads()
            self._graph2.replay()
        self.num_updates += 1
        K.ffn_exchange_poll()  # (outside the graph: a 4-byte copy + an event; examined at a later update, never stalls)
        return self._graph_out
