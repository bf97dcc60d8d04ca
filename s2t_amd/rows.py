"""Packed rows: the frames of a batch without its padding (include/s2t_hip.h, "Packed rows").

The reference pads every utterance of a batch to the longest one (data/audio/speech_to_text_dataset.py:411-485) and runs
every row-wise module on the padded frames (models/speech_to_text/s2t_transformer.py:1765-1946).  Here an activation keeps
its padded SIZE, ``[B * T, C]``, but utterance ``b`` occupies rows ``cu[b] .. cu[b] + lens[b] - 1``, followed by its halo
rows — the at most ``halo`` frames behind its end (never beyond ``T``) whose depthwise-convolution output is not zero and
enters the BatchNorm batch statistics of the reference (modules/convolution.py:94-104, statistics over all B * T frames).
Halo rows are treated like padded frames (masked, no gradient); the padded frames behind them — zero in the reference after
its masks, with zero gradient — are not stored.  Rows at and beyond ``cu[B]`` are never read or written: the live row count
is read on the device, so one captured hipGraph serves batches of any fill.

The geometry hangs on the batch's int32 lengths tensor as ``lens._pk`` (the same tensor every layer already passes for its
masks); ``s2t_amd.kernels`` turns it into the row map / ``cu`` arguments of the C-ABI.
"""
import os

import torch

from . import functional as Fn
from . import kernels as K

# S2T_PACKED=0 keeps the padded layout everywhere (A/B measurements; the fp32 parity mode always does)
ENABLED = os.environ.get("S2T_PACKED", "1") != "0"
# Row counts from which a batch runs packed: below them the row-block kernels leave most of the chip idle either way and the
# geometry launch is not paid back.  Module attributes so that a test can run the packed kernels on an oracle-sized batch.
MIN_ENC_ROWS = int(os.environ.get("S2T_PACKED_MIN_ENC_ROWS", "4096"))  # B * T' of an encoder (or PDS stage)
MIN_DEC_ROWS = int(os.environ.get("S2T_PACKED_MIN_DEC_ROWS", "2048"))  # B * U target rows of a decoder


class PackedRows:
    """Device-resident geometry of one packed batch: ``cu`` [B + 1] int32 and the row map (header + one int32 per row:
    ``(b << 16) | t`` on rows that hold a frame, -1 on halo rows and beyond the live rows; ``map[-1]`` = live rows)."""

    HEADER = 4

    def __init__(self, lens32, B, T, halo, tag=None):
        assert lens32.dtype == torch.int32 and lens32.is_cuda and T <= 65535 and B <= 32767
        self.lens, self.B, self.T, self.halo = lens32, int(B), int(T), int(halo)
        self.M = self.B * self.T
        # ``tag`` names the user of the geometry (an encoder, a decoder, a PDS stage): its memo entry is then REPLACED when the next
        # batch object comes along (an entry per lengths tensor would grow without bound over an eager epoch)
        self.cu, self.buf = Fn.batch_memo(("packed_rows", tag if tag is not None else Fn.memo_owner(lens32), self.B, self.T, self.halo),
                                          (lens32,), self._build)
        self.map_ptr = self.buf.data_ptr() + 4 * self.HEADER

    def _build(self, lens):
        B, T, M = self.B, self.T, self.M
        if B <= 1024:  # one launch (s2t_rows_geometry); the torch composition below is its restatement for larger batches
            # (a refresh — new lengths copied into the same tensor, Trainer.load_batch — rewrites the tensors a captured step reads)
            cu = getattr(self, "cu", None)
            buf = getattr(self, "buf", None)
            if cu is None:
                cu = torch.empty(B + 1, dtype=torch.int32, device=lens.device)
                buf = torch.empty(self.HEADER + M, dtype=torch.int32, device=lens.device)
            K.rows_geometry(lens, B, T, self.halo, cu, buf)
            return cu, buf
        cap = lens + (T - lens).clamp(min=0, max=self.halo)
        cu = torch.zeros(B + 1, dtype=torch.int32, device=lens.device)
        cu[1:] = torch.cumsum(cap, 0)
        r = torch.arange(M, dtype=torch.int32, device=lens.device)
        b = torch.searchsorted(cu[1:].contiguous(), r, right=True)
        bc = b.clamp(max=B - 1)
        t = r - cu[bc.long()]
        valid = (b < B) & (t < lens[bc.long()])
        buf = torch.empty(self.HEADER + M, dtype=torch.int32, device=lens.device)
        buf[:self.HEADER] = 0
        buf[self.HEADER - 1] = cu[B]
        buf[self.HEADER:] = torch.where(valid, (bc.to(torch.int32) << 16) | t, torch.full_like(t, -1))
        return cu, buf

    @property
    def row_map(self):
        return self.buf[self.HEADER:]

    def live_rows(self):
        """Host copy of the live row count (tests and tools only: a device-to-host copy)."""
        return int(self.buf[self.HEADER - 1])


class LazyList(list):
    """A list of tensors produced on first access: ``LazyList([thunk, ...])`` (``len`` does not materialise)."""

    def __init__(self, thunks):
        super().__init__(thunks)
        self._done = [False] * len(thunks)

    def _get(self, i):
        if not self._done[i]:
            super().__setitem__(i, super().__getitem__(i)())
            self._done[i] = True
        return super().__getitem__(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self)))]
        return self._get(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self)))


def attach(lens32, B, T, halo, tag=None):
    """Give ``lens32`` the packed geometry of its batch (idempotent per (B, T, halo)); returns ``lens32``."""
    g = K.rows_geom(lens32)
    if g is None or (g.B, g.T, g.halo) != (B, T, halo):
        lens32._pk = PackedRows(lens32, B, T, halo, tag)
    return lens32


def detached(lens32):
    """The same lengths without the geometry (a view sharing the storage): for the kernels that stay on padded rows."""
    return lens32.view(-1) if K.rows_geom(lens32) is not None else lens32


class PackFn(torch.autograd.Function):
    """padded [B*T, C] -> packed rows (halo rows zero); backward scatters the gradient back (padded frames: zero)."""

    @staticmethod
    def forward(ctx, x, lens):
        out = torch.empty_like(x)
        K.pack_rows(x.contiguous(), out, lens, True)
        ctx.lens = lens
        return out

    @staticmethod
    def backward(ctx, dy):
        dx = torch.zeros_like(dy)
        K.pack_rows(dy.contiguous(), dx, ctx.lens, False)
        return dx, None


class UnpackFn(torch.autograd.Function):
    """packed rows -> padded [B*T, C] with zero padded frames (what the reference's masked tensors hold)."""

    @staticmethod
    def forward(ctx, x, lens):
        out = torch.zeros_like(x)
        K.pack_rows(x.contiguous(), out, lens, False)
        ctx.lens = lens
        return out

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty_like(dy)
        K.pack_rows(dy.contiguous(), dx, ctx.lens, True)
        # rows beyond the live ones are not written by the kernel and never read by a packed consumer
        return dx, None


def pack(x, lens):
    return PackFn.apply(x, lens)


def unpack(x, lens):
    return UnpackFn.apply(x, lens)
