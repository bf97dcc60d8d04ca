"""Flat parameter storage sized for 288 GB of HBM3E: ONE contiguous fp32 master buffer, ONE fp32 gradient
buffer and (bf16 mode) ONE bf16 shadow buffer for the whole model.

Why (MI355X-first, not the reference's layout):
  * the optimizer, the gradient norm and the fp32->bf16 weight refresh are single launches over the buffer;
  * data-parallel gradient all-reduce works on contiguous slices of the gradient buffer (buckets are views,
    no copy-in / copy-out as in the reference's LegacyDistributedDataParallel, legacy_distributed_data_parallel.py:82-120);
  * parameters that are consumed together (q/k/v projection weights) are placed adjacently so one GEMM reads them
    as a single [3d, d] matrix while the state_dict keeps the reference's per-tensor keys.
"""
from typing import Dict, List

import torch
import torch.nn as nn

ALIGN = 64  # elements; keeps every tensor 128-byte aligned in the bf16 shadow (256 B in fp32)


class FlatParameters:
    def __init__(self, module: nn.Module, compute_dtype: torch.dtype = torch.float32):
        self.module = module
        self.compute_dtype = compute_dtype
        params: List[nn.Parameter] = []
        seen = set()

        def add(p):
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)

        # adjacency groups first (in declaration order), then everything else in registration order
        for m in module.modules():
            groups = getattr(m, "flat_groups", None)
            if groups is not None:
                for grp in groups():
                    for p in grp:
                        add(p)
        for p in module.parameters():
            add(p)
        self.params = params
        self.offsets: Dict[int, int] = {}
        off = 0
        groups_flat = set()
        for m in module.modules():
            groups = getattr(m, "flat_groups", None)
            if groups is not None:
                for grp in groups():
                    for p in grp[:-1]:
                        groups_flat.add(id(p))
        for p in params:
            self.offsets[id(p)] = off
            n = p.numel()
            # members of a group (except the last) must be followed immediately by the next member
            if id(p) in groups_flat:
                assert n % 8 == 0, "grouped parameters must keep 16-byte alignment"
                off += n
            else:
                off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        dev = params[0].device
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        for p in params:
            o, n = self.offsets[id(p)], p.numel()
            self.master[o:o + n].copy_(p.data.reshape(-1).float())
            p.data = self.master[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._s2t_shadow = self.shadow[o:o + n].view(p.shape) if self.shadow is not None else None
            p._s2t_flat = self
        # transposed bf16 copies of 2-D weights for kernels that want them that way (s2t_ffn_fused_bwd): registered on first
        # use, ALL refreshed by one launch at most once per backward pass (see transposed())
        self._wt = {"params": [], "bufs": {}, "table": None, "fresh": False, "ext": (0, 0)}
        self.refresh_shadow()

    def refresh_shadow(self):
        """fp32 master -> bf16 shadow (after loading a checkpoint; Adam refreshes it itself each step)."""
        if self.shadow is not None:
            if self.master.is_cuda:
                from . import kernels as K

                K.cast_f32_to_bf16(self.master, self.shadow, self.numel)
            else:
                self.shadow.copy_(self.master)

    def mark_transposed_stale(self):
        """The transposed bf16 weight copies (``transposed``) no longer match the shadow: rewrite them at their next use."""
        self._wt["fresh"] = False

    def zero_grad(self):
        self.grad.zero_()

    def reattach_grads(self):
        """``p.grad`` views of the flat gradient buffer for every parameter whose ``.grad`` was dropped (fairseq's optimizers
        zero gradients by setting them to None, optim/fairseq_optimizer.py:129-133); returns True when any was missing —
        the flat buffer is then zeroed, which is what that zero_grad meant."""
        missing = [p for p in self.params if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[id(p)]]
        if not missing:
            return False
        self.grad.zero_()
        for p in self.params:
            o, n = self.offsets[id(p)], p.numel()
            p.grad = self.grad[o:o + n].view(p.shape)
        return True

    def view(self, first: nn.Parameter, rows: int, cols: int, what: str = "compute") -> torch.Tensor:
        """A [rows, cols] matrix starting at ``first`` (spanning an adjacency group)."""
        o = self.offsets[id(first)]
        buf = {"compute": self.shadow if self.shadow is not None else self.master, "master": self.master,
               "grad": self.grad}[what]
        return buf[o:o + rows * cols].view(rows, cols)


def transposed(p: nn.Parameter, in_backward: bool, rows: int = 0, cols: int = 0) -> torch.Tensor:
    """bf16 [cols, rows] copy of the 2-D parameter ``p`` (bf16 mode only), current with the shadow.  The copies of every
    registered parameter are rewritten by ONE s2t_transpose_bf16_batched launch, the first time any of them is asked for in a
    backward pass (``in_backward``: the caller has the end-of-backward callback armed, which marks them stale again);
    outside a backward pass every call refreshes.  ``rows`` / ``cols``: the [rows, cols] matrix STARTING at ``p`` in the flat
    buffer instead (an adjacency group, e.g. the fused [3d, d] q | k | v projection; a Conv1d(k=1) weight as [out, in])."""
    from . import kernels as K

    flat = p._s2t_flat
    st = flat._wt
    key = (id(p), rows, cols)
    new = key not in st["bufs"]
    if new:
        assert flat.shadow is not None
        src = flat.view(p, rows, cols) if rows else p._s2t_shadow
        assert src.dim() == 2
        st["bufs"][key] = torch.empty(src.shape[1], src.shape[0], dtype=torch.bfloat16, device=flat.shadow.device)
        st["params"].append((key, src))
        st["table"] = None
    if new or not st["fresh"] or not in_backward:
        if st["table"] is None:
            import numpy as np

            rec = np.zeros(len(st["params"]), dtype=np.dtype([("src", "u8"), ("dst", "u8"), ("rows", "i4"), ("cols", "i4")]))
            for i, (kq, src) in enumerate(st["params"]):
                rec[i] = (src.data_ptr(), st["bufs"][kq].data_ptr(), src.shape[0], src.shape[1])
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("s2t_amd: a transposed weight copy was first asked for during graph capture; run one "
                                   "eager step first")
            st["table"] = torch.from_numpy(rec.view(np.uint8).copy()).to(flat.shadow.device)
            tl = K.transpose_tiles([tuple(src.shape) for _, src in st["params"]])
            st["ext"] = (torch.from_numpy(tl).to(flat.shadow.device), int(tl.shape[0]))
        K.transpose_batched(st["table"], len(st["params"]), st["ext"][0], st["ext"][1])
        st["fresh"] = in_backward
    return st["bufs"][key]


def cw(p: nn.Parameter) -> torch.Tensor:
    """Compute-dtype view of a parameter (bf16 shadow in bf16 mode, the fp32 master otherwise)."""
    s = getattr(p, "_s2t_shadow", None)
    return s if s is not None else p.data
