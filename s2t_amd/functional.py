"""Autograd functions of the S2T hot path: each forward/backward is a short sequence of libs2t_hip kernels.

Everything here is batch-major: activations are row matrices ``[B*T, C]`` (row = b*T + t).  Weight
gradients are ACCUMULATED in place into the flat fp32 gradient buffer (``p.grad`` are views of it, see
flat_params.py), so ``backward`` returns ``None`` for parameters; this is what lets the data-parallel
wrapper all-reduce contiguous buckets while the rest of backward is still running.

There is no CPU path: every function raises if its inputs are not on the GPU.
"""
import math
import os

import torch

from . import kernels as K
from .flat_params import cw, transposed

_HOOKS = {"grad_ready": None}  # set by the DDP wrapper: called with a parameter once its gradient is final


def _ready(*params):
    cb = _HOOKS["grad_ready"]
    if cb is not None and _BE["armed"]:  # gradient work is still queued for the end of backward: notify after it ran
        _BE["ready"].extend(p for p in params if p is not None)
        return
    if cb is not None:
        for p in params:
            if p is not None:
                cb(p)


def _pad8(n):
    return (n + 7) // 8 * 8


class _DropoutState:
    """Seed (device-resident, so a captured hipGraph replays with fresh masks) + a per-step site counter.  Every
    dropout site of a step gets a unique id at forward time and hands it to its backward, which regenerates the mask."""

    def __init__(self):
        self.seed = None
        self.site = 0

    def begin_step(self, device):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if self.seed is None or self.seed.device != dev:
            self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.site = 0

    def set_seed(self, value):
        self.seed.fill_(int(value))

    def next(self, p, device):
        """(p, seed tensor, site id) for an active site, None when p == 0."""
        if not p or p <= 0.0:
            return None
        if self.seed is None:
            self.begin_step(device)
        self.site += 1
        return (float(p), self.seed, self.site)


DROPOUT = _DropoutState()


_FUSE_LN_DROP = os.environ.get("S2T_FUSE_LN_DROP", "1") != "0"
DROP_STATS = {"handed_over": 0, "launched": 0}  # how the dropped branch gradients of backward passes were obtained


def _same_drop(a, b):
    return a is not None and b is not None and a[0] == b[0] and a[2] == b[2] and a[1].data_ptr() == b[1].data_ptr()


def _tag_drop(y, drop):
    """Mark a block output with the mask of its output dropout: the LayerNorm that consumes it can then produce the
    dropped branch gradient in its own backward pass (LayerNormFn), instead of a separate dropout launch."""
    if drop is not None and _FUSE_LN_DROP and y is not None:
        y._s2t_drop_o = drop
    return y


def _hand_over(dx, drop, dxd):
    """Leave dropout(dx) under ``drop`` beside dx for the backward of the block in front (picked up by _drop_rows)."""
    dx._s2t_dropped = (drop, dxd, dx.data_ptr(), dx._version)


def _drop_rows(x, drop):
    """x * mask / (1-p) for a [rows, cols] matrix (mask convention of the GEMM epilogue); identity when drop is None."""
    if drop is None:
        return x
    ready = getattr(x, "_s2t_dropped", None)  # left there by the LayerNorm backward that produced x
    # The hand-over only stands while x still IS the buffer that LayerNorm backward wrote: when the tagged activation has
    # a second consumer the autograd engine accumulates the other branch's gradient into it IN PLACE (old.add_(new); the
    # Python object and its attribute survive), which bumps the version counter — then the dropped copy is stale.
    if (ready is not None and _same_drop(ready[0], drop) and ready[1].shape == x.shape
            and ready[2] == x.data_ptr() and ready[3] == x._version):
        DROP_STATS["handed_over"] += 1
        return ready[1]
    DROP_STATS["launched"] += 1
    rows, cols = x.shape
    out = torch.empty(rows, cols, dtype=x.dtype, device=x.device)
    K.dropout(x, x.stride(0), out, cols, rows, cols, drop)
    return out


class DropoutFn(torch.autograd.Function):
    """FairseqDropout (modules/fairseq_dropout.py) on a row matrix."""

    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return _drop_rows(x.contiguous(), drop)

    @staticmethod
    def backward(ctx, dy):
        return _drop_rows(dy.contiguous(), ctx.drop), None


def dropout(x, p, training):
    drop = DROPOUT.next(p if training else 0.0, x.device)
    return x if drop is None else DropoutFn.apply(x, drop)


def fused(params, rows, cols):
    """[rows, cols] compute-dtype view over adjacent parameters (flat_params adjacency group)."""
    first = cw(params[0])
    exp = first.data_ptr()
    for p in params:
        t = cw(p)
        assert t.data_ptr() == exp, "parameters are not adjacent in the flat buffer (call flatten first)"
        exp += t.numel() * t.element_size()
    return first.as_strided((rows, cols), (cols, 1))


def fused_grad(params, rows, cols):
    first = params[0].grad
    return first.as_strided((rows, cols), (cols, 1))


def fused_master(params, n):
    return params[0].data.as_strided((n,), (1,))


# Weight gradients are off the critical path of backward (nothing consumes them before the optimizer), so they CAN run
# on a second HIP stream, forked behind an event when their operands are ready and joined at the end of the backward
# pass (an autograd engine callback); their operands are kept alive until the join so the caching allocator cannot hand
# the memory to main-stream kernels.  Opt-in (S2T_WGRAD_STREAM=1): measured on MI355X it LOSES 5 % (24.6 vs 23.4 ms per
# step) — two persistent GEMM grids competing for the two 64-KiB-LDS workgroup slots of each CU and for L2 cost more
# than the idle CU time of the critical-path kernels they were meant to fill.
_WG = {"stream": None, "keep": [], "armed": False, "enabled": os.environ.get("S2T_WGRAD_STREAM", "0") == "1"}


def wgrad_stream():
    """The side stream weight-gradient kernels are queued on (None when the feature is off or unused so far)."""
    return _WG["stream"]


def _wgrad_join():
    s = _WG["stream"]
    if s is not None:
        torch.cuda.current_stream().wait_stream(s)
    _WG["keep"].clear()
    _WG["armed"] = False


class _on_wgrad_stream:
    """Context manager: run the enclosed launches on the weight-gradient stream, ordered after everything queued on the
    current stream so far; ``keep`` tensors stay referenced until the end-of-backward join."""

    def __init__(self, *keep):
        self.keep = keep
        self.ctx = None

    def __enter__(self):
        if not _WG["enabled"] or not torch.cuda.is_available():
            return self
        if _WG["stream"] is None:
            _WG["stream"] = torch.cuda.Stream()
        side = _WG["stream"]
        side.wait_stream(torch.cuda.current_stream())
        _WG["keep"].append(self.keep)
        if not _WG["armed"]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join)
            except RuntimeError:  # not inside a backward pass: stay on the current stream
                _WG["keep"].pop()
                return self
            _WG["armed"] = True
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# Work deferred to the end of the backward pass (an autograd engine callback): the fold of every LayerNorm's
# per-replica dgamma/dbeta partial sums (one launch instead of one per LayerNorm) and the grouped weight gradients.
# Per-batch bookkeeping that depends only on a batch's LENGTHS / TOKEN IDS (subsampled lengths, padding masks, decoder
# positions, CTC target matrices): a handful of tiny sort / scan / compare launches each.  batch_memo computes them once per
# batch object (keyed by the identity and version of the source tensors) instead of once per forward pass, so a captured step
# replays none of them; when new data is copied INTO a captured step's static batch (Trainer.load_batch),
# refresh_batch_memos recomputes every memo into the SAME output tensors (the graph holds their addresses).
_MEMO = {}          # key -> (srcs, versions, fn, outs): the most recent batch object of each user
_MEMO_PINNED = {}   # key -> [entries whose tensors a captured hipGraph reads]: never replaced or released while pinned
_UID = [0]


_UID_SELF = {}  # uid -> weak reference to the object the serial was made for


def memo_owner(obj):
    """A process-unique serial for ``obj`` (an encoder, a decoder, a criterion, a trainer) to key its memos by.  NOT ``id(obj)``:
    CPython re-uses the id of a collected object, so a model built after another was dropped could land on the dead one's keys (and,
    with a captured step, on tensors whose addresses that step's graph had baked in).  When ``obj`` is collected its entries go.
    The serial rides in the object's ``__dict__``, which ``copy.deepcopy`` / pickling carry over to the copy (an EMA model): a
    serial found on an object it was not made for is replaced by a fresh one, so two live owners never share keys (and the
    original's finalizer never purges the copy's entries)."""
    import weakref

    uid = getattr(obj, "_s2t_uid", None)
    if uid is not None:
        ref = _UID_SELF.get(uid)
        if ref is not None and ref() is obj:
            return uid
        uid = None  # inherited through a copy (or its first owner is gone): this object gets its own
    _UID[0] += 1
    uid = ("uid", _UID[0])
    try:
        object.__setattr__(obj, "_s2t_uid", uid)
        _UID_SELF[uid] = weakref.ref(obj)
        weakref.finalize(obj, _purge_owner, uid)
    except (AttributeError, TypeError):  # an object without attributes / weak references: the serial cannot stay with it
        _UID_SELF.pop(uid, None)
        return ("id", id(obj))
    return uid


def _key_has(key, uid):
    if key == uid:
        return True
    return isinstance(key, tuple) and any(_key_has(k, uid) for k in key)


def _purge_owner(uid):
    _UID_SELF.pop(uid, None)
    for table in (_MEMO, _MEMO_PINNED):
        for key in [k for k in table if _key_has(k, uid)]:
            table.pop(key, None)


def _same_srcs(e, srcs):
    return len(e[0]) == len(srcs) and all(a is b for a, b in zip(e[0], srcs))


def _recompute_in_place(table_entry, key, pinned_slot=None):
    """fn(*srcs) into the entry's EXISTING output tensors (addresses a captured graph may hold stay valid)."""
    srcs, _, fn, outs = table_entry
    into = getattr(fn, "into", None)
    if into is not None and all((not torch.is_tensor(t)) or t.is_cuda for t in tuple(srcs) + tuple(outs)):
        # the memo's own one-launch form (csrc/bookkeeping.hip) writes straight into the existing tensors: no temporaries, no
        # copies (composed from ATen ops the refresh of a step's bookkeeping was 45 launches, 0.23 ms of the headline step)
        into(*srcs, outs=outs)
        return (srcs, tuple(t._version for t in srcs), fn, outs)
    new = fn(*srcs)
    ok = len(new) == len(outs) and all((not torch.is_tensor(o)) or (torch.is_tensor(n) and o.shape == n.shape and o.dtype == n.dtype)
                                       for o, n in zip(outs, new))
    if not ok:
        return None
    for o, n in zip(outs, new):
        if torch.is_tensor(o) and o is not n:
            o.copy_(n)
    return (srcs, tuple(t._version for t in srcs), fn, outs)


def batch_memo(key, srcs, fn):
    """``fn(*srcs)`` computed once per batch object.  Same source OBJECTS with new contents (their version counters moved: data
    copied into a static batch) are recomputed INTO the existing outputs, so whoever holds those tensors — a later layer of this
    pass, a captured graph — sees the new values at the old addresses; another batch object replaces the entry, unless a captured
    graph reads it (pin_batch_memos): the pinned entry then stays beside the new one.

    A moved version refreshes through ``refresh_batch_memos``: every memo computed from the moved sources AND every memo computed
    from the outputs of one refreshed here (the padding mask feeds the length memos, the lengths the packed-row geometry) is
    recomputed in the same sweep.  The one-launch ``into`` forms write through raw addresses, so an output's own version counter
    never moves: without the cascade a dependent memo would pass its version check and hand back the previous batch's values
    (ADVICE round 5)."""
    for table in (_MEMO_PINNED.get(key, ()), None):
        if table is None:
            e = _MEMO.get(key)
        else:
            e = next((x for x in table if _same_srcs(x, srcs)), None)
        if e is None or not _same_srcs(e, srcs):
            continue
        if all(a._version == v for a, v in zip(e[0], e[1])):
            return e[3]
        refresh_batch_memos([a for a, v in zip(e[0], e[1]) if a._version != v])
        if table is None:
            e = _MEMO.get(key)  # (dropped by the refresh when its outputs changed shape: computed afresh below)
            if e is not None and _same_srcs(e, srcs):
                return e[3]
        else:
            return next(x for x in table if _same_srcs(x, srcs))[3]
    outs = fn(*srcs)
    _MEMO[key] = (tuple(srcs), tuple(t._version for t in srcs), fn, outs)
    return outs


def pin_batch_memos(tensors, owner):
    """Called when a step over the batch ``tensors`` is captured into a hipGraph: every memo computed from them (directly or
    through another memo's outputs) is read by the graph at its current address — from now on it is neither replaced by an eager
    pass over another batch nor released, until ``unpin_batch_memos(owner)``."""
    reach = {id(t) for t in tensors}
    moved = True
    while moved:
        moved = False
        for key, e in list(_MEMO.items()):
            if any(id(t) in reach for t in e[0]):
                _MEMO.pop(key)
                _MEMO_PINNED.setdefault(key, []).append(e + (owner,))
                for o in e[3]:
                    if torch.is_tensor(o):
                        reach.add(id(o))
                moved = True
    # (entries keep their 4-tuple shape for the readers above; the owner rides as a fifth element)


def unpin_batch_memos(owner):
    """Release ``owner``'s pins.  An entry whose key has no newer entry in the table goes BACK there (as the most recent batch of
    its user) instead of being dropped: a re-capture over the same static batch then finds — and pins — the very tensors its warm-up
    steps read, where dropping them left the bookkeeping to be recomputed inside the stream capture, in the graph's private
    pool, unpinned."""
    for key in list(_MEMO_PINNED):
        keep = []
        for e in _MEMO_PINNED[key]:
            if len(e) > 4 and e[4] == owner:
                if key not in _MEMO:
                    _MEMO[key] = tuple(e[:4])
            else:
                keep.append(e)
        if keep:
            _MEMO_PINNED[key] = keep
        else:
            _MEMO_PINNED.pop(key)


def _all_memo_entries():
    for key, e in list(_MEMO.items()):
        yield _MEMO, key, None, e
    for key, lst in list(_MEMO_PINNED.items()):
        for i, e in enumerate(lst):
            yield lst, key, i, e


def refresh_batch_memos(changed):
    """``changed``: the tensors whose contents were just overwritten.  Every memo computed from one of them — or from the
    output of a memo refreshed here (the padding mask feeds the length memos) — is recomputed into its existing tensors."""
    dirty = {id(t) for t in changed}
    # Passes until nothing new becomes dirty: a memo may sit in the table BEFORE one it depends on (a key first made for an
    # earlier batch keeps its place when a later batch replaces its entry, while that batch's own geometry memos are appended
    # behind it), so one sweep in insertion order can recompute it from a source that is refreshed only afterwards.
    seen = {}  # (key, slot) -> the dirty sources it was last recomputed from
    progress = True
    while progress:
        progress = False
        for table, key, slot, e in _all_memo_entries():
            srcs, outs = e[0], e[3]
            d = frozenset(id(t) for t in srcs if id(t) in dirty)
            if not d or seen.get((key, slot)) == d:
                continue
            ne = _recompute_in_place(e[:4], key)
            if ne is None:
                if slot is None:
                    # an unpinned entry whose outputs changed shape (e.g. a longer longest target): nobody holds its addresses
                    # but this pass — drop it; batch_memo computes it afresh, and memos made from its old outputs no longer match
                    _MEMO.pop(key, None)
                    progress = True
                    continue
                raise RuntimeError("s2t_amd: the per-batch bookkeeping %r of a captured step changed shape under a refresh" % (key,))
            for o in outs:
                if torch.is_tensor(o):
                    dirty.add(id(o))
            if slot is None:
                _MEMO[key] = ne
            else:
                table[slot] = ne + tuple(e[4:])
            seen[(key, slot)] = d
            progress = True


_BE = {"armed": False, "ready": [], "flats": []}
_LNQ = {"entries": [], "pools": {}, "off": {}}


def _arm_backward_end():
    """True when running inside a backward pass with the end-of-backward callback installed."""
    if not _BE["armed"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_on_backward_end)
            _BE["armed"] = True
        except RuntimeError:  # not inside a backward pass
            return False
    return True


def _ln_workspace(cols, device):
    """A zeroed [replicas][2][cols] slice of the per-device pool (the fold kernel leaves it zeroed again)."""
    key = str(device)
    n = K.LN_REPLICAS * 2 * cols
    pool = _LNQ["pools"].get(key)
    off = _LNQ["off"].get(key, 0)
    if pool is None or off + n > pool.numel():
        pool = torch.zeros(max(1 << 22, 2 * (off + n)), dtype=torch.float32, device=device)
        _LNQ["pools"][key] = pool  # slices of the previous pool stay alive through the queued entries
        off = 0
    _LNQ["off"][key] = off + n
    return pool[off:off + n]


_DBD = {}


def _dbd_buffer(H, B, Tq, ldB, dt, dev):
    """The [H, B, Tq, ldB] buffer of skewed dS shared by all layers of a shape: zero-filled once — every backward call
    overwrites exactly the band of each row, the rest stays zero (s2t_attn_fused_bwd, dbd_band_only), and a layer's
    consumers (the two position GEMMs) run on the same stream before the next layer's kernel writes it again."""
    key = (str(dev), dt, H, B, Tq, ldB)
    buf = _DBD.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("s2t_amd: a new attention-gradient shape appeared during graph capture; run one eager step first")
        # a few shapes at most (PDS stages, bucketed lengths); the oldest goes — unless a captured graph may hold its address
        # (every buffer that existed when a capture was recorded stays for the life of the process: GRAPH_PINNED)
        while len(_DBD) >= 8:
            victim = next((k for k in _DBD if k not in GRAPH_PINNED), None)
            if victim is None:
                break
            _DBD.pop(victim)
        buf = _DBD[key] = torch.zeros(H, B, Tq, ldB, dtype=dt, device=dev)
    return buf


GRAPH_PINNED = set()


def pin_buffers_for_graph():
    """Called when a step is captured into a hipGraph: buffers allocated OUTSIDE the graph's pool whose addresses the graph
    bakes in (the shared skewed-dS buffers) may never be freed or re-used for another shape afterwards."""
    GRAPH_PINNED.update(_DBD.keys())
    K.graphs_hold_workspaces()


_POSQ = {"entries": [], "pool": {}, "next": {}, "parts": []}
_POSQ_CAP = 32
# Deferred folds: the per-utterance partial tables of the position-table gradient and the partial rows of the depthwise weight
# gradient stay in per-layer scratch and are summed at the end of backward, all layers in ONE launch each (a 5 us launch per
# layer otherwise).  Slots beyond the cap fall back to the immediate fold.
_FOLD_DEFER = os.environ.get("S2T_FOLD_DEFER", "1") != "0"
_FOLD_CAP = 24
_DWQ = {"entries": []}


def _posq_slot(n_pos, d, dev, zero):
    """A [n_pos, d] fp32 slice for one layer's position-table gradient.  Slices are handed out from the END of a pooled
    block, so that the layers' slices ascend in memory like their linear_pos weight gradients do (backward visits the
    layers last to first) and one batched GEMM can walk both with positive strides."""
    key = (str(dev), n_pos, d)
    pool = _POSQ["pool"].get(key)
    if pool is None:
        pool = _POSQ["pool"][key] = torch.zeros(_POSQ_CAP, n_pos, d, dtype=torch.float32, device=dev)
    k = _POSQ["next"].get(key, _POSQ_CAP) - 1
    if k < 0:
        return (torch.zeros if zero else torch.empty)(n_pos, d, dtype=torch.float32, device=dev)
    _POSQ["next"][key] = k
    sl = pool[k]
    if zero:
        sl.zero_()
    return sl


def _flush_posq():
    """linear_pos weight gradients (espnet_multihead_attention.py:331 backward): dW_l += dp_l^T pos_tab for every queued
    layer, as ONE batched GEMM per run of layers whose slices and gradients sit at constant strides."""
    entries, _POSQ["entries"] = _POSQ["entries"], []
    _POSQ["next"] = {}
    parts, _POSQ["parts"] = _POSQ["parts"], []
    groups = {}
    for part, dp, dims in parts:  # the layers' partial tables -> their dp slices, one launch per shape
        groups.setdefault(dims, ([], []))
        groups[dims][0].append(part)
        groups[dims][1].append(dp)
    for (B, H, Tq, dk), (ps, ds) in groups.items():
        K.relpos_dp_reduce(ps, ds, B, H, Tq, dk)
    if not entries:
        return
    entries.sort(key=lambda e: e[2].data_ptr())
    i = 0
    while i < len(entries):
        dp, pos32, g, n_pos, d = entries[i]
        j = i + 1
        gs = ds = None
        while j < len(entries):
            e = entries[j]
            if e[1] is not pos32 or e[3] != n_pos or e[4] != d:
                break
            gs_j = (e[2].data_ptr() - entries[j - 1][2].data_ptr()) // 4
            ds_j = (e[0].data_ptr() - entries[j - 1][0].data_ptr()) // 4
            if gs is None:
                gs, ds = gs_j, ds_j
            if gs_j != gs or ds_j != ds or gs < d * d or ds < n_pos * d or gs % 4 or ds % 4:
                break
            j += 1
        L = j - i
        K.gemm(dp, pos32, g, M=d, N=d, K=n_pos, lda=d, ldb=d, ldc=d, a_kmajor=True, b_kmajor=True, batch=L,
               a_s=(ds or 0, 0), b_s=(0, 0), c_s=(gs or 0, 0),
               split_k=_POSW_SPLIT if _POSW_SPLIT else max(1, min(8, (n_pos + 63) // 64)), c_atomic=True)
        i = j


def _flush_deferred():
    """Run the gradient work queued so far (LayerNorm parameter folds, grouped weight gradients) and deliver the
    grad-ready notifications that were waiting for it."""
    entries, _LNQ["entries"] = _LNQ["entries"], []
    if entries:
        K.layernorm_fold(entries)
    _flush_posq()
    dwq, _DWQ["entries"] = _DWQ["entries"], []
    groups = {}
    for ws, dw, rows, n in dwq:  # depthwise weight-gradient partial rows -> the gradients, one launch per shape
        groups.setdefault((rows, n), ([], []))
        groups[(rows, n)][0].append(ws)
        groups[(rows, n)][1].append(dw)
    for (rows, n), (wss, dws) in groups.items():
        K.rows_fold_add(wss, dws, rows, n)
    flush_wgrads()
    ready, _BE["ready"] = _BE["ready"], []
    cb = _HOOKS["grad_ready"]
    if cb is not None:
        for p in ready:
            cb(p)


def _on_backward_end():
    _flush_deferred()
    for flat in _BE["flats"]:
        flat._wt["fresh"] = False  # the optimizer may change the weights before the next backward pass
    _BE["flats"] = []
    for k in _LNQ["off"]:
        _LNQ["off"][k] = 0
    _BE["armed"] = False
    _BE["count"] = _BE.get("count", 0) + 1


def backward_count():
    """Backward passes through this package's autograd Functions completed so far (an optimizer step follows one)."""
    return _BE.get("count", 0)


class GradStageFn(torch.autograd.Function):
    """Identity in forward.  In backward it closes a STAGE: the weight gradients queued by everything behind this point
    (decoder, upper encoder layers, ...) run now instead of at the end of the pass, so that their parameters are final
    while the rest of backward is still ahead — which is what lets the data-parallel wrapper start reducing those buckets
    beside the remaining backward kernels (the reference reduces after backward: legacy_distributed_data_parallel.py:76-160).
    Only active when a grad-ready hook is installed (a data-parallel step): a single-GPU step keeps one grouped launch."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        if _HOOKS["grad_ready"] is not None and _BE["armed"]:
            _flush_deferred()
        return dy


def grad_stage(x):
    return GradStageFn.apply(x) if (torch.is_grad_enabled() and x.requires_grad) else x


# Number of gradient stages of a data-parallel backward pass: 4 = behind the decoder and at each third of the encoder stack,
# 3 (default) = decoder + upper third | middle third | lower third + front-end, 2 = one cut a third of the stack from the
# bottom, 1 = none (one grouped launch, every bucket reduced after backward).  More stages start the all-reduce earlier but
# each grouped launch is less efficient than one (single-rank data-parallel step on the headline configuration: 14.83 /
# 14.56 / 14.48 / 14.24 ms for 4 / 3 / 2 / 1 stages, same-box alternation); with 60 MB of bf16 gradients per step the last stage's buckets are
# what stays exposed.
GRAD_STAGES = int(os.environ.get("S2T_GRAD_STAGES", "3"))


def grad_stage_layers(n):
    """Encoder layer indices in front of which a gradient stage closes in backward (see GradStageFn)."""
    if n < 6 or GRAD_STAGES <= 1:
        return set()
    if GRAD_STAGES == 2:
        return {n // 3}
    return {n // 3, (2 * n) // 3}


# Deferred, grouped weight gradients (csrc/gemm_grouped.hip): nothing consumes a weight gradient before the optimizer,
# so the bf16 (dY, X, dW) triples of a backward pass are queued and executed by ONE persistent launch (+ one reduction)
# from an autograd engine callback at the end of backward.  Operands stay referenced until then.  The parameters'
# grad-ready notifications (DDP bucket hooks) are deferred with them.  S2T_WGRAD_GROUPED=0 restores one GEMM per weight.
# Policy (S2T_WGRAD_GROUPED): "graph" (default) = only while a hipGraph is being captured — in eager mode the host
# cannot enqueue the (now much shorter) backward kernels fast enough once the weight-gradient launches are gone, the
# GPU starves and a step takes 34-48 ms instead of 23; "1" = always, "0" = never.
_WGQ = {"probs": [], "mode": os.environ.get("S2T_WGRAD_GROUPED", "graph"),
        "bufs": {}, "captured": []}


def reserve_wgrad_staging(device, nbytes=8 << 20, count=1):
    """Allocate ``count`` pinned/device staging pairs ahead of a graph capture (nothing may be allocated from the host
    allocator while a stream is capturing; every flush of a captured step dedicates one pair to the graph)."""
    key = str(torch.device(device) if not isinstance(device, torch.device) else device)
    ring = _WGQ["bufs"].setdefault(key, {"slots": [], "i": 0})
    have = sum(1 for sl in ring["slots"] if sl[0].numel() >= nbytes)
    for _ in range(max(0, count - have)):
        ring["slots"].append([torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                              torch.empty(nbytes, dtype=torch.uint8, device=device), None])


_WG_KSTEPS = int(os.environ.get("S2T_WG_KSTEPS", "64"))  # K-steps (of 64 rows) per work item
_WG_256 = os.environ.get("S2T_WG_256", "1") != "0"          # s2t_wgrad_grouped256 where the operands allow it
_WG_KSTEPS256 = int(os.environ.get("S2T_WG_KSTEPS256", "256"))  # its K-steps (of 32 rows) per work item, about (splits are balanced)
_WG_AUTO = os.environ.get("S2T_WG_AUTO", "1") != "0"  # choose them per launch from the item count (flush_wgrads)
_WG_DTYPE = None


def _wg_dtype():
    global _WG_DTYPE
    if _WG_DTYPE is None:
        import numpy as np
        _WG_DTYPE = np.dtype([("A", "u8"), ("B", "u8"), ("C", "u8"), ("colsum", "u8"), ("lda", "i8"), ("ldb", "i8"),
                              ("ldc", "i8"), ("ws_base", "i8"), ("M", "i4"), ("N", "i4"), ("K", "i4"), ("tiles_n", "i4"),
                              ("ksteps", "i4"), ("nsplit", "i4"), ("alpha", "f4"), ("next", "i4"), ("k_live", "u8")])
        assert _WG_DTYPE.itemsize == 104
    return _WG_DTYPE


def wgrad256_eligible(M, ldy, ldx, py=0, px=0):
    """Operand layout rules of s2t_wgrad_grouped256 (LDS-DMA of 16-byte pieces, 32-bit byte offsets inside an operand)."""
    return (_WG_256 and ldy % 8 == 0 and ldx % 8 == 0 and py % 16 == 0 and px % 16 == 0
            and M * ldy * 2 < 2 ** 31 and M * ldx * 2 < 2 ** 31)


def flush_wgrads():
    """Run every queued weight gradient (called by the end-of-backward callback; safe to call with an empty queue).
    The queue is split by kernel: the problems that meet the operand layout rules of the 256 x 256 LDS-DMA kernel run on it in
    one launch, the others (an unaligned leading dimension, an operand beyond 2 GiB) on the 128 x 128 register-staged kernel in a
    second one — one odd problem no longer moves the whole backward pass to the slow kernel, or aborts a packed batch."""
    q, _WGQ["probs"] = _WGQ["probs"], []
    q = [e if len(e) == 11 else tuple(e) + (None,) for e in q]  # (dY, X, dW, Nout, Kin, M, ldy, ldx, alpha, db, packed geometry)
    if not q:
        return
    ok = [wgrad256_eligible(M, ldy, ldx, dY.data_ptr(), X.data_ptr()) for (dY, X, _, _, _, M, ldy, ldx, _, _, _) in q]
    if any(e[10] is not None and not k for e, k in zip(q, ok)):
        # (S2TTransformerEncoder._packed_ok and the decoder's gate keep such shapes on padded rows: wgrad256_eligible)
        raise RuntimeError("s2t_amd: a weight gradient over packed rows does not meet the operand layout rules of the 256 x 256 "
                           "grouped kernel, the only one that reads the live row count on the device")
    # tied weights (one dW, several problems) are chained inside ONE launch: a chain stays together, on the 128 x 128 kernel if
    # any of its members needs it
    slow_dw = {e[2].data_ptr() for e, k in zip(q, ok) if not k}
    fast = [e for e, k in zip(q, ok) if k and e[2].data_ptr() not in slow_dw]
    slow = [e for e, k in zip(q, ok) if not (k and e[2].data_ptr() not in slow_dw)]
    if any(e[10] is not None for e in slow):
        raise RuntimeError("s2t_amd: a packed weight gradient is tied to one that needs the 128 x 128 grouped kernel")
    if fast:
        _flush_wgrad_group(fast, True)
    if slow:
        _flush_wgrad_group(slow, False)


def _flush_wgrad_group(q, big):
    import numpy as np
    if q:
        dev = q[0][0].device
        probs = np.zeros(len(q), dtype=_wg_dtype())
        items, tiles = [], []
        ws_floats = 0
        k_tail = False
        last_of = {}
        TL, KS, per_item = (256, 32, _WG_KSTEPS256) if big else (128, 64, _WG_KSTEPS)
        if big and _WG_AUTO:
            # K-steps per work item chosen per launch: the items of a launch run in rounds of one workgroup per CU (128 KiB of
            # LDS each), so a data-parallel gradient stage with ~300 items of 250 steps runs two rounds, the second a fifth
            # full, where 468 items of 167 steps fill both.  Estimated time = rounds x (longest item + the cost of its
            # partial tile, ~60 steps).
            slots = torch.cuda.get_device_properties(dev).multi_processor_count
            best = None
            for cand in (256, 200, 170, 128, 100, 64):
                n_items, longest = 0, 0
                for (_, _, _, Nout, Kin, M, _, _, _, _, _) in q:
                    kt = (M + KS - 1) // KS
                    ns = max(1, (kt + cand // 2) // cand)
                    n_items += ((Nout + TL - 1) // TL) * ((Kin + TL - 1) // TL) * ns
                    longest = max(longest, (kt + ns - 1) // ns)
                cost = ((n_items + slots - 1) // slots) * (longest + 60)
                if best is None or cost < best[0] * 0.97:  # ties and near-ties go to the larger items (less workspace traffic)
                    best = (cost, cand)
            per_item = best[1]
        for i, (dY, X, dW, Nout, Kin, M, ldy, ldx, alpha, db, live) in enumerate(q):
            tm_n, tn_n = (Nout + TL - 1) // TL, (Kin + TL - 1) // TL
            ktiles = (M + KS - 1) // KS
            nsplit = (ktiles + per_item - 1) // per_item
            if big:  # balanced splits of about `per_item` K-steps (500 steps: 2 x 250, not 256 + 244 or 3 x 192 with a short one)
                nsplit = max(1, (ktiles + per_item // 2) // per_item)
                per_item_p = (ktiles + nsplit - 1) // nsplit
            else:
                per_item_p = per_item
            k_tail |= (M % 64) != 0
            probs[i] = (dY.data_ptr(), X.data_ptr(), dW.data_ptr(), db.data_ptr() if db is not None else 0, ldy, ldx,
                        Kin, ws_floats, Nout, Kin, M, tn_n, per_item_p, nsplit, alpha, -1, live.map_ptr - 4 if live is not None else 0)
            prev = last_of.get(dW.data_ptr())  # tied weights: chain the problems, reduce them in one workgroup
            last_of[dW.data_ptr()] = i
            if prev is not None:
                assert probs[prev]["M"] == Nout and probs[prev]["N"] == Kin
                probs[prev]["next"] = i
            ws_floats += tm_n * tn_n * nsplit * TL * TL
            # order: K split, tile row, tile column — the ~64 items an XCD works on at a time then read the SAME K-range
            # of both operands (its 4 MiB L2 serves the re-reads; measured 15 GB -> fabric traffic per launch before)
            t = np.empty((nsplit, tm_n, tn_n, 4), dtype=np.int32)
            t[..., 0] = i
            t[..., 1] = np.arange(tm_n, dtype=np.int32)[None, :, None]
            t[..., 2] = np.arange(tn_n, dtype=np.int32)[None, None, :]
            t[..., 3] = np.arange(nsplit, dtype=np.int32)[:, None, None]
            items.append(t.reshape(-1, 4))
            if prev is not None:
                continue  # its tiles are reduced by the head of the chain
            tl = np.zeros((tm_n, tn_n, 4), dtype=np.int32)
            tl[..., 0] = i
            tl[..., 1] = np.arange(tm_n, dtype=np.int32)[:, None]
            tl[..., 2] = np.arange(tn_n, dtype=np.int32)[None, :]
            tiles.append(tl.reshape(-1, 4))
        items = np.concatenate(items)
        tiles = np.concatenate(tiles)
        blob = np.concatenate([probs.view(np.uint8), items.view(np.uint8).reshape(-1), tiles.view(np.uint8).reshape(-1)])
        nbytes = (blob.size + 15) // 16 * 16
        # Staging: the tables are written into pinned memory by the CPU and copied to the device by the stream.  The CPU
        # runs ahead of the GPU, so a staging block may only be rewritten once the copy AND the kernels that read the
        # device copy have executed: a ring of blocks, each guarded by an event (host-side wait only on wrap-around).
        capturing = torch.cuda.is_current_stream_capturing()
        key = str(dev)
        ring = _WGQ["bufs"].setdefault(key, {"slots": [], "i": 0})
        cap = max(nbytes, 1 << 20)
        if capturing:
            # a captured graph replays the host-to-device copy from ITS pinned block on every replay: take a block the
            # eager warm-up steps allocated (no allocation while capturing) and dedicate it to the graph
            fit = [sl for sl in ring["slots"] if sl[0].numel() >= nbytes]
            if fit:
                slot = fit[0]
                ring["slots"].remove(slot)  # its last eager use has completed: capture() synchronises before capturing
            else:
                slot = [torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=dev), None]
            _WGQ["captured"].append(slot)
            bufs = (slot[0], slot[1])
            slot_ev = None
        else:
            if len(ring["slots"]) < 4:
                ring["slots"].append([torch.empty(cap, dtype=torch.uint8).pin_memory(),
                                      torch.empty(cap, dtype=torch.uint8, device=dev), None])
                slot = ring["slots"][-1]
            else:
                ring["i"] = (ring["i"] + 1) % len(ring["slots"])
                slot = ring["slots"][ring["i"]]
                if slot[2] is not None:
                    slot[2].synchronize()
                if slot[0].numel() < nbytes:
                    slot[0] = torch.empty(cap, dtype=torch.uint8).pin_memory()
                    slot[1] = torch.empty(cap, dtype=torch.uint8, device=dev)
            bufs = (slot[0], slot[1])
            slot_ev = slot
        host, devb = bufs
        host[:blob.size].copy_(torch.from_numpy(blob))
        devb[:nbytes].copy_(host[:nbytes], non_blocking=True)
        o1 = probs.nbytes
        o2 = o1 + items.nbytes
        ws = K._scratch("wgrad_grouped", ws_floats, dev)
        if big:
            K.wgrad_grouped256(devb, len(q), devb[o1:], items.shape[0], devb[o2:], tiles.shape[0], ws)
        else:
            K.wgrad_grouped(devb, len(q), devb[o1:], items.shape[0], devb[o2:], tiles.shape[0], ws, k_tail)
        if slot_ev is not None:
            ev = torch.cuda.Event()
            ev.record()
            slot_ev[2] = ev


def _wgrad(dY, X, dW, Nout, Kin, M, ldy, ldx, alpha=1.0, db=None, rows=None):
    """dW[Nout, Kin] += alpha * dY[M, Nout]^T @ X[M, Kin]   (TN GEMM, split-K over M, two-phase workspace reduction);
    db[Nout] += alpha * column sums of dY when given (taken from the staged dY tiles inside the same kernel).
    bf16 problems inside a backward pass are queued for the grouped launch (flush_wgrads)."""
    mode = _WGQ["mode"]
    live = K.rows_geom(rows)
    if live is not None:
        # packed batch: M (the reduction dimension) is the padded bound, the live row count is read on the device — only the
        # grouped launch does that (s2t_wgrad_problem.k_live), inside and outside a captured step
        if not (dY.dtype == torch.bfloat16 and _arm_backward_end()):
            raise RuntimeError("s2t_amd: weight gradients over packed rows run through the grouped launch (bf16, inside backward)")
        _WGQ["probs"].append((dY, X, dW, Nout, Kin, M, ldy, ldx, float(alpha), db, live))
        return
    if dY.dtype == torch.bfloat16 and dY.is_cuda and (mode == "1" or (mode == "graph" and torch.cuda.is_current_stream_capturing())):
        if _arm_backward_end():
            _WGQ["probs"].append((dY, X, dW, Nout, Kin, M, ldy, ldx, float(alpha), db, None))
            return
    tiles = ((Nout + 127) // 128) * ((Kin + 127) // 128)
    bke = 64 if dY.dtype == torch.bfloat16 else 32
    ktiles = (M + bke - 1) // bke
    # about 2 workgroups per CU in total, but never fewer than 8 K-steps per workgroup: storing and re-reading a
    # 128x128 fp32 partial tile costs about as much as a few K-steps (measured: tools/wgrad_bench.py)
    split = max(1, min((ktiles + 7) // 8, (512 + tiles - 1) // tiles))
    with _on_wgrad_stream(dY, X):
        K.gemm(dY, X, dW, M=Nout, N=Kin, K=M, lda=ldy, ldb=ldx, ldc=Kin, a_kmajor=True, b_kmajor=True, alpha=alpha,
               split_k=split, c_atomic=True, colsum_a=db)


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    """modules/layer_norm.py:30-35; optional fused padded-row mask on the output.
    ``fork``: also return x itself for the residual branch of a pre-LN block (x + f(LN(x))); the two gradients that
    meet at x are then added inside the LayerNorm backward kernel instead of by a separate elementwise pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, lens, T, fork, bound=None):
        rows, cols = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        K.layernorm_fwd(x, gamma.data, beta.data, y, mean, rstd, rows, cols, 1e-5, lens, T, bound=bound)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta, ctx.lens, ctx.T, ctx.bound = gamma, beta, lens, T, bound
        up = getattr(x, "_s2t_drop_o", None)  # output-dropout mask of the block that produced x (see _tag_drop)
        ctx.up_drop = up if (up is not None and x.dtype == torch.bfloat16 and cols == 256) else None
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, mean, rstd = ctx.saved_tensors
        rows, cols = x.shape
        dx = torch.empty_like(x)
        if dres is not None:
            dres = dres.contiguous()
        dxd = torch.empty_like(x) if ctx.up_drop is not None else None
        if x.is_cuda and _arm_backward_end():
            # partial sums into a private workspace slice; ONE fold launch for all LayerNorms at the end of backward
            ws = _ln_workspace(cols, x.device)
            K.layernorm_bwd(x, ctx.gamma.data, dy.contiguous(), mean, rstd, dx, None, None, rows, cols, ctx.lens, ctx.T,
                            dres, ws=ws, dx_drop=dxd, drop=ctx.up_drop, bound=ctx.bound)
            _LNQ["entries"].append((ws, ctx.gamma.grad, ctx.beta.grad, cols))
        else:
            K.layernorm_bwd(x, ctx.gamma.data, dy.contiguous(), mean, rstd, dx, ctx.gamma.grad, ctx.beta.grad, rows, cols,
                            ctx.lens, ctx.T, dres, dx_drop=dxd, drop=ctx.up_drop, bound=ctx.bound)
        _ready(ctx.gamma, ctx.beta)
        if dxd is not None:
            _hand_over(dx, ctx.up_drop, dxd)
        return dx, None, None, None, None, None, None


def layer_norm(x, gamma, beta, lens=None, T=0, fork=False, rows=None):
    """``rows``: the lengths tensor of a packed batch (s2t_amd/rows.py) when there is no mask: only its live rows are
    normalised (the same keyword on the other row-wise functions of this module)."""
    return LayerNormFn.apply(x, gamma, beta, lens, T, fork, rows)


def _ln_backward(x, gamma, beta, dy, mean, rstd, lens, T, dres, up_drop, rows=None):
    """s2t_layernorm_bwd for a LayerNorm whose forward was folded into a row-block kernel: parameter gradients through the
    queued fold (or directly outside a backward pass), the residual-branch gradient ``dres`` added in the kernel, and the
    dropped copy for the block in front handed over when ``up_drop`` names its output mask."""
    nrows, cols = x.shape
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if up_drop is not None else None
    if x.is_cuda and _arm_backward_end():
        ws = _ln_workspace(cols, x.device)
        K.layernorm_bwd(x, gamma.data, dy, mean, rstd, dx, None, None, nrows, cols, lens, T, dres, ws=ws, dx_drop=dxd, drop=up_drop,
                        bound=rows)
        _LNQ["entries"].append((ws, gamma.grad, beta.grad, cols))
    else:
        K.layernorm_bwd(x, gamma.data, dy, mean, rstd, dx, gamma.grad, beta.grad, nrows, cols, lens, T, dres, dx_drop=dxd,
                        drop=up_drop, bound=rows)
    _ready(gamma, beta)
    if dxd is not None:
        _hand_over(dx, up_drop, dxd)
    return dx


_RB = os.environ.get("S2T_ROWBLOCK", "1") != "0"          # row-block projection kernels (csrc/rowblock.hip)
_RB_MIN_ROWS = int(os.environ.get("S2T_ROWBLOCK_MIN_ROWS", "2048"))  # 64-row blocks: fewer rows leave most CUs idle (the decoder's ~3900 rows still gain: one launch instead of LayerNorm + GEMM)


def _rb_ok(x, N, act=None):
    return _RB and K.rowblock_supported(x, N, act) and x.shape[0] >= _RB_MIN_ROWS


_DGRAD_SPLITK = os.environ.get("S2T_DGRAD_SPLITK", "1") != "0"
_DGRAD_SPLITK_MINK = int(os.environ.get("S2T_DGRAD_SPLITK_MINK", "1024"))


_RB_DGRAD = os.environ.get("S2T_RB_DGRAD", "1") != "0"  # s2t_rowblock_dgrad: projection dgrad + LayerNorm backward in one launch


def _dgrad_ln_backward(dy, first_w, Kd, x_pre, gamma, beta, mean, rstd, lens, T, dres, up_drop, rows=None):
    """dx = LayerNorm'(dy @ W) + dres through s2t_rowblock_dgrad, or None when it does not apply (the caller then runs the
    GEMM and s2t_layernorm_bwd).  ``first_w``: first parameter of the [Kd, 256] weight (group) in the flat buffer."""
    flat = getattr(first_w, "_s2t_flat", None)
    M = dy.shape[0]
    if not (_RB_DGRAD and flat is not None and flat.shadow is not None and dy.dtype == torch.bfloat16 and dy.is_contiguous()
            and x_pre.shape[1] == 256 and Kd % 256 == 0 and Kd <= 2048 and M >= _RB_MIN_ROWS and (M + 64) * Kd * 2 < 2 ** 32
            and x_pre.is_contiguous() and _arm_backward_end()):
        return None
    if flat not in _BE["flats"]:
        _BE["flats"].append(flat)
    wt = transposed(first_w, True, Kd, 256)
    dx = torch.empty_like(x_pre)
    dxd = torch.empty_like(x_pre) if up_drop is not None else None
    ws = _ln_workspace(256, x_pre.device)
    K.rowblock_dgrad(dy, wt, ln=dict(x=x_pre, gamma=gamma.data, mean=mean, rstd=rstd, ws=ws, dx=dx, dres=dres, lens=lens, T=T,
                                     dx_drop=dxd, drop=up_drop), rows=rows)
    _LNQ["entries"].append((ws, gamma.grad, beta.grad, 256))
    _ready(gamma, beta)
    if dxd is not None:
        _hand_over(dx, up_drop, dxd)
    return dx


_RB_DGRAD_PLAIN = os.environ.get("S2T_RB_DGRAD_PLAIN", "1") != "0"
_BN_IN_PW2 = os.environ.get("S2T_BN_IN_PW2", "1") != "0"  # training BatchNorm apply + activation in pointwise conv 2's prologue


def _dgrad_rowblock(dy, w_param, Kd, rows=None):
    """dx[M, 256] = dy[M, Kd] @ W for a [Kd, 256] weight (attention output projection, pointwise conv 2: Kd = 256) through
    the row-block dgrad kernel without a LayerNorm behind it, or None when it does not apply (the caller runs s2t_gemm)."""
    flat = getattr(w_param, "_s2t_flat", None)
    M = dy.shape[0]
    if not (_RB_DGRAD_PLAIN and _RB_DGRAD and flat is not None and flat.shadow is not None and dy.dtype == torch.bfloat16
            and dy.is_contiguous() and dy.shape[1] == Kd and Kd % 256 == 0 and Kd <= 2048 and M >= _RB_MIN_ROWS
            and (M + 64) * Kd * 2 < 2 ** 32 and _arm_backward_end()):
        return None
    if flat not in _BE["flats"]:
        _BE["flats"].append(flat)
    wt = transposed(w_param, True, Kd, 256)
    dx = torch.empty(M, 256, dtype=dy.dtype, device=dy.device)
    K.rowblock_dgrad(dy, wt, dxn=dx, rows=rows)
    return dx


def _dgrad(dy, w, dx, M, N, Kd, lda, ldb, ldc, alpha=1.0, rows=None):
    """dx[M, N] = alpha * dy[M, Kd] @ w[Kd, N] (input gradient of a linear layer).  A long reduction over few output tiles —
    the vocabulary projections (K = V = 10 000; M = B*U decoder rows: 32 tiles walking 157 K-steps each took 180 us) and the
    decoder's FFN — is cut into K splits whose fp32 partial tiles meet in a workspace (two-phase split-K, bf16 result)."""
    split = 1
    if _DGRAD_SPLITK and dy.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and Kd >= _DGRAD_SPLITK_MINK:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        if tiles < 384:
            split = max(1, min(8, 512 // tiles, Kd // 256))
            if K.gemm_configure() != 0 and Kd % 8 == 0:
                # the large tiles of gemm256.hip take (tile, split) pairs as work items: about one round of 256-row tiles where a
                # split of at least eight K-steps allows it (16000 x 256 x 10000: 173 us on 128 x 128 tiles with 2 splits, 119 on
                # 256-row tiles with 4; tools/splitk_probe.py) — else s2t_gemm falls back to 128-row tiles or the 128 x 128 kernel
                t256 = ((M + 255) // 256) * ((N + 255) // 256)
                s256 = min(8, -(-230 // t256))
                if s256 > 1 and t256 * s256 >= 150 and Kd // (64 * s256) >= 24:  # (shorter splits: 128-row tiles at the split above
                    split = s256                                                  # measured better, 16000 x 256 x 3072: 42 against 46 us)
    if split > 1:
        K.gemm(dy, w, dx, M=M, N=N, K=Kd, lda=lda, ldb=ldb, ldc=ldc, b_kmajor=True, alpha=alpha, split_k=split, c_atomic=2,
               rows=rows)
    else:
        K.gemm(dy, w, dx, M=M, N=N, K=Kd, lda=lda, ldb=ldb, ldc=ldc, b_kmajor=True, alpha=alpha, rows=rows)


# ------------------------------------------------------------------------------------------------
# Linear (+ residual)
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = residual + alpha * (x @ W^T + b)   (F.linear; out_dtype lets logits come out in fp32).
    The output buffer's row stride is padded to a multiple of 8 elements (16-byte row starts for any
    vocabulary size); the returned tensor is the [:, :N] view of it."""

    @staticmethod
    def forward(ctx, x, w, b, alpha, residual, out_dtype, rows=None):
        M, Kin = x.shape
        Nout = w.shape[0]
        ldc = _pad8(Nout)
        y = torch.empty(M, ldc, dtype=out_dtype or x.dtype, device=x.device)
        ldr = residual.stride(0) if residual is not None else 0
        K.gemm(x, cw(w), y, M=M, N=Nout, K=Kin, lda=Kin, ldb=Kin, ldc=ldc, bias=b.data if b is not None else None,
               alpha=alpha, residual=residual, ldr=ldr, rows=rows)
        ctx.save_for_backward(x)
        ctx.w, ctx.b, ctx.alpha, ctx.has_res, ctx.rows = w, b, alpha, residual is not None, rows
        return y[:, :Nout] if ldc != Nout else y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        M, Kin = x.shape
        Nout = w.shape[0]
        ld = _pad8(Nout)
        if dy.dtype != x.dtype or dy.stride(1) != 1 or dy.stride(0) % 8 or dy.data_ptr() % 16:
            buf = torch.zeros(M, ld, dtype=x.dtype, device=x.device) if ld != Nout else torch.empty(M, ld, dtype=x.dtype, device=x.device)
            buf[:, :Nout].copy_(dy)
            dy = buf[:, :Nout]
        ldy = dy.stride(0)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _dgrad(dy, cw(w), dx, M, Kin, Nout, ldy, Kin, Kin, ctx.alpha, rows=ctx.rows)
        _wgrad(dy, x, w.grad, Nout, Kin, M, ldy, Kin, ctx.alpha, b.grad if b is not None else None, rows=ctx.rows)
        _ready(w, b)
        return dx, None, None, None, (dy if ctx.has_res else None), None, None


def linear(x, w, b=None, alpha=1.0, residual=None, out_dtype=None, rows=None):
    return LinearFn.apply(x, w, b, alpha, residual, out_dtype, rows)


# ------------------------------------------------------------------------------------------------
# Feed-forward block
# ------------------------------------------------------------------------------------------------
class FFNFn(torch.autograd.Function):
    """out = residual + alpha * (W2 act(W1 x + b1) + b2)   (modules/s2t_transformer_layer.py:55-66, :258-265, :311-317;
    decoder FFN modules/transformer_layer.py:520-530)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, alpha, residual, train, drop_h, drop_o, rows=None):
        M, d = x.shape
        F_ = w1.shape[0]
        h = torch.empty(M, F_, dtype=x.dtype, device=x.device)
        z = torch.empty(M, F_, dtype=x.dtype, device=x.device) if train else None
        ctx.rows = rows  # packed batch (s2t_amd/rows.py): only the live rows are computed
        K.gemm(x, cw(w1), h, M=M, N=F_, K=d, lda=d, ldb=d, ldc=F_, bias=b1.data, act=act, preact=z, ldp=F_, drop=drop_h, rows=rows)
        y = torch.empty(M, d, dtype=x.dtype, device=x.device)
        # the decoder's rows (B*U = 3904: 62 output tiles walking 32 K-steps each) leave three CUs in four idle: the long
        # reduction is cut into splits and the epilogue runs in the second phase
        split = 1
        if _FWD_SPLITK and x.dtype == torch.bfloat16 and F_ >= 1024 and rows is None:
            tiles = ((M + 127) // 128) * ((d + 127) // 128)
            if tiles < 128:
                split = max(1, min(_FWD_SPLITK, 512 // tiles, F_ // 256))
        K.gemm(h, cw(w2), y, M=M, N=d, K=F_, lda=F_, ldb=F_, ldc=d, bias=b2.data, alpha=alpha, residual=residual, ldr=d,
               drop=drop_o, split_k=split, c_atomic=2 if split > 1 else False, rows=rows)
        ctx.drops = (drop_h, drop_o)
        if train:
            ctx.save_for_backward(x, z, h)
        ctx.p = (w1, b1, w2, b2)
        ctx.act, ctx.alpha = act, alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, h = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.p
        M, d = x.shape
        F_ = w1.shape[0]
        dres = dy.contiguous()
        drop_h, drop_o = ctx.drops
        dy = _drop_rows(dres, drop_o)  # gradient of the branch behind the output dropout
        # dZ = alpha * dropout_h(dY @ W2) * act'(Z)
        dz = torch.empty(M, F_, dtype=x.dtype, device=x.device)
        rows = ctx.rows
        K.gemm(dy, cw(w2), dz, M=M, N=F_, K=d, lda=d, ldb=F_, ldc=F_, b_kmajor=True, alpha=ctx.alpha, dact_z=z, ldz=F_,
               dact=ctx.act, drop=drop_h, rows=rows)
        _wgrad(dy, h, w2.grad, d, F_, M, d, F_, ctx.alpha, b2.grad, rows=rows)
        _ready(w2, b2)
        _wgrad(dz, x, w1.grad, F_, d, M, F_, d, 1.0, b1.grad, rows=rows)
        dx = torch.empty_like(x)
        _dgrad(dz, cw(w1), dx, M, d, F_, F_, d, d, rows=rows)
        _ready(w1, b1)
        return dx, None, None, None, None, None, None, dres, None, None, None, None


_FFN_FUSED = os.environ.get("S2T_FFN_FUSED", "1") != "0"
_FWD_SPLITK = int(os.environ.get("S2T_FWD_SPLITK", "8"))  # most K splits of the (unfused) FFN's second forward GEMM; 0: one pass
_FFN_FUSED_BWD = os.environ.get("S2T_FFN_FUSED_BWD", "1") != "0"  # s2t_ffn_fused_bwd for the block's input gradient
_FFN_FUSED_MIN_ROWS = int(os.environ.get("S2T_FFN_FUSED_MIN_ROWS", "1024"))  # (8192 before the eight-way split of csrc/ffn_pc.hip)


class FFNBlockFn(torch.autograd.Function):
    """One launch for a whole pre-LN feed-forward block (csrc/rowblock.hip, s2t_ffn_fused_fwd):
         y = x + alpha * drop_o(W2 drop_h(act(W1 LN(x) + b1)) + b2)      [-> LN_end(y), padded rows zeroed]
    (modules/s2t_transformer_layer.py:55-66, :258-265, :311-320).  The backward pass is the unfused one (dgrad GEMMs,
    grouped weight gradients, s2t_layernorm_bwd with the residual gradient folded in) on the tensors the kernel saved.
    Returns LN_end(y) when ``end`` = (gamma, beta) is given (the pre-norm y has no other consumer), y otherwise."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, act, alpha, train, drop_h, drop_o, end_g, end_b, end_lens, end_T,
                rows=None):
        M, d = x.shape
        F_ = w1.shape[0]
        dev = x.device
        bf = torch.bfloat16
        y = torch.empty(M, d, dtype=bf, device=dev) if (train or end_g is None) else None
        y_ln = torch.empty(M, d, dtype=bf, device=dev) if end_g is not None else None
        x_ln = mean = rstd = z = h = emean = erstd = None
        if train:
            x_ln = torch.empty(M, d, dtype=bf, device=dev)
            mean = torch.empty(M, dtype=torch.float32, device=dev)
            rstd = torch.empty(M, dtype=torch.float32, device=dev)
            z = torch.empty(K.ffn_z_rows(M), F_, dtype=bf, device=dev)  # either layout fits (tiled: row blocks of 128)
            h = torch.empty(M, F_, dtype=bf, device=dev)
            if end_g is not None:
                emean = torch.empty(M, dtype=torch.float32, device=dev)
                erstd = torch.empty(M, dtype=torch.float32, device=dev)
        # z may come back in the 128-row kernel's TILED layout when the backward pass is the fused kernel too (its only reader)
        fused_bwd_ok = (_FFN_FUSED_BWD and getattr(w1, "_s2t_flat", None) is not None and w1._s2t_flat.shadow is not None
                        and getattr(w2, "_s2t_flat", None) is w1._s2t_flat and (M + 128) * F_ * 2 < 2 ** 32)
        ctx.z_tiled = K.ffn_fused_fwd(x, cw(w1), b1.data, cw(w2), b2.data, y, act=act, alpha=alpha, residual=x,
                                      ln=(gamma.data, beta.data), end_ln=(end_g.data, end_b.data) if end_g is not None else None,
                                      y_ln=y_ln, end_stats=(emean, erstd) if emean is not None else None, end_lens=end_lens,
                                      end_T=end_T, x_ln=x_ln, ln_stats=(mean, rstd) if train else None, z=z, h=h, drop_h=drop_h,
                                      drop_o=drop_o, z_tiled_ok=train and fused_bwd_ok, rows=rows)
        if train:
            ctx.save_for_backward(x, x_ln, mean, rstd, z, h, y if end_g is not None else None, emean, erstd)
        ctx.p = (gamma, beta, w1, b1, w2, b2, end_g, end_b)
        ctx.act, ctx.alpha, ctx.drops, ctx.end_lens, ctx.end_T = act, alpha, (drop_h, drop_o), end_lens, end_T
        ctx.rows = rows if rows is not None else (end_lens if K.rows_geom(end_lens) is not None else None)
        up = getattr(x, "_s2t_drop_o", None)  # output-dropout mask of the block that produced x (see _tag_drop)
        ctx.up_drop = up
        return y_ln if end_g is not None else y

    @staticmethod
    def backward(ctx, dout):
        x, x_ln, mean, rstd, z, h, y, emean, erstd = ctx.saved_tensors
        gamma, beta, w1, b1, w2, b2, end_g, end_b = ctx.p
        M, d = x.shape
        F_ = w1.shape[0]
        drop_h, drop_o = ctx.drops
        dout = dout.contiguous()
        queued = x.is_cuda and _arm_backward_end()
        rows = ctx.rows

        def ln_bwd(xx, g_, b_, dy_, mean_, rstd_, lens, T, dres, dx_drop, drop):
            dx_ = torch.empty_like(xx)
            if queued:
                ws = _ln_workspace(d, xx.device)
                K.layernorm_bwd(xx, g_.data, dy_, mean_, rstd_, dx_, None, None, M, d, lens, T, dres, ws=ws, dx_drop=dx_drop,
                                drop=drop, bound=rows)
                _LNQ["entries"].append((ws, g_.grad, b_.grad, d))
            else:
                K.layernorm_bwd(xx, g_.data, dy_, mean_, rstd_, dx_, g_.grad, b_.grad, M, d, lens, T, dres, dx_drop=dx_drop,
                                drop=drop, bound=rows)
            _ready(g_, b_)
            return dx_

        fused_bwd = (_FFN_FUSED_BWD and getattr(w1, "_s2t_flat", None) is not None and w1._s2t_flat.shadow is not None
                     and getattr(w2, "_s2t_flat", None) is w1._s2t_flat and M * F_ * 2 < 2 ** 32)
        assert fused_bwd or not ctx.z_tiled, "a tiled z is read by the fused backward kernel only"
        zt = bool(ctx.z_tiled)
        if not zt:
            z = z[:M]
        end_in_kernel = None
        if end_g is not None and fused_bwd and queued:
            # the trailing LayerNorm's backward rides in the fused kernel's prologue: it writes dres (gradient w.r.t. y) and
            # dropout(dres) under this block's own output mask (the products' input and W2's weight-gradient operand)
            dres = torch.empty_like(x)
            dy = torch.empty_like(x) if drop_o is not None else dres
            ws_e = _ln_workspace(d, x.device)
            end_in_kernel = dict(y=y, gamma=end_g.data, mean=emean, rstd=erstd, ws=ws_e, dres=dres, lens=ctx.end_lens,
                                 T=ctx.end_T, dy=dy if drop_o is not None else None, drop=drop_o)
            _LNQ["entries"].append((ws_e, end_g.grad, end_b.grad, d))
            _ready(end_g, end_b)
        elif end_g is not None:
            # gradient of the trailing LayerNorm w.r.t. y; dropout(dy) under this block's own output mask comes with it
            dyd = torch.empty_like(x) if drop_o is not None else None
            dres = ln_bwd(y, end_g, end_b, dout, emean, erstd, ctx.end_lens, ctx.end_T, None, dyd, drop_o)
            dy = dyd if dyd is not None else dres
        else:
            dres = dout
            dy = _drop_rows(dres, drop_o)
        # dZ = alpha * dropout_h(dY @ W2) * act'(Z), dXn = dZ @ W1
        dz = torch.empty(M, F_, dtype=x.dtype, device=x.device)
        dxd = torch.empty_like(x) if ctx.up_drop is not None else None
        dx = None
        if fused_bwd:
            if queued and w1._s2t_flat not in _BE["flats"]:
                _BE["flats"].append(w1._s2t_flat)
            w2t, w1t = transposed(w2, queued), transposed(w1, queued)
            if queued:  # the leading LayerNorm's backward rides in the kernel's epilogue (parameter sums through the fold)
                dx = torch.empty_like(x)
                ws = _ln_workspace(d, x.device)
                K.ffn_fused_bwd(dout if end_in_kernel is not None else dy, w2t, w1t, z, dz, None, act=ctx.act, alpha=ctx.alpha,
                                drop_h=drop_h, end=end_in_kernel, z_tiled=zt, rows=rows,
                                ln=dict(x=x, gamma=gamma.data, mean=mean, rstd=rstd, ws=ws, dx=dx, dres=dres, dx_drop=dxd,
                                        drop=ctx.up_drop))
                _LNQ["entries"].append((ws, gamma.grad, beta.grad, d))
                _ready(gamma, beta)
            else:
                dxl = torch.empty_like(x)
                K.ffn_fused_bwd(dy, w2t, w1t, z, dz, dxl, act=ctx.act, alpha=ctx.alpha, drop_h=drop_h, z_tiled=zt, rows=rows)
        else:
            dxl = torch.empty_like(x)
            K.gemm(dy, cw(w2), dz, M=M, N=F_, K=d, lda=d, ldb=F_, ldc=F_, b_kmajor=True, alpha=ctx.alpha, dact_z=z, ldz=F_,
                   dact=ctx.act, drop=drop_h, rows=rows)
            _dgrad(dz, cw(w1), dxl, M, d, F_, F_, d, d, rows=rows)
        _wgrad(dy, h, w2.grad, d, F_, M, d, F_, ctx.alpha, b2.grad, rows=rows)
        _ready(w2, b2)
        _wgrad(dz, x_ln, w1.grad, F_, d, M, F_, d, 1.0, b1.grad, rows=rows)
        _ready(w1, b1)
        if dx is None:
            dx = ln_bwd(x, gamma, beta, dxl, mean, rstd, None, 0, dres, dxd, ctx.up_drop)
        if dxd is not None:
            _hand_over(dx, ctx.up_drop, dxd)
        return (dx,) + (None,) * 16


def ffn_block(x, norm_g, norm_b, w1, b1, w2, b2, act, alpha, p_hidden=0.0, p_out=0.0, training=False, end_norm=None,
              end_lens=None, end_T=0, rows=None):
    """Pre-LN feed-forward block with its residual (and the layer's trailing LayerNorm when ``end_norm`` = (gamma, beta)):
    the row-block kernel when it applies, the LayerNorm / GEMM composition otherwise."""
    M, d = x.shape
    F_ = w1.shape[0]
    if (_FFN_FUSED and K.ffn_fused_supported(x, F_) and M >= _FFN_FUSED_MIN_ROWS and x.is_contiguous()
            and act in ("relu", "swish")):
        drop_h = DROPOUT.next(p_hidden if training else 0.0, x.device)
        drop_o = DROPOUT.next(p_out if training else 0.0, x.device)
        eg, eb = end_norm if end_norm is not None else (None, None)
        out = FFNBlockFn.apply(x, norm_g, norm_b, w1, b1, w2, b2, act, alpha, torch.is_grad_enabled(), drop_h, drop_o, eg,
                               eb, end_lens, end_T, rows)
        return out if end_norm is not None else _tag_drop(out, drop_o)
    # the LayerNorm / GEMM composition (other widths, fp32); packed rows (s2t_amd/rows.py) pass through as the live-row bound
    pk = rows if K.rows_geom(rows) is not None else (end_lens if K.rows_geom(end_lens) is not None else None)
    y, xr = layer_norm(x, norm_g, norm_b, fork=True, rows=pk)
    out = ffn(y, w1, b1, w2, b2, act, alpha, xr, p_hidden, p_out, training, rows=pk)
    if end_norm is not None:
        out = layer_norm(out, end_norm[0], end_norm[1], end_lens, end_T, rows=pk if end_lens is None else None)
    return out


def ffn(x, w1, b1, w2, b2, act, alpha, residual, p_hidden=0.0, p_out=0.0, training=False, rows=None):
    drop_h = DROPOUT.next(p_hidden if training else 0.0, x.device)
    drop_o = DROPOUT.next(p_out if training else 0.0, x.device)
    return _tag_drop(FFNFn.apply(x, w1, b1, w2, b2, act, alpha, residual, torch.is_grad_enabled(), drop_h, drop_o,
                                 rows if K.rows_geom(rows) is not None else None), drop_o)


# ------------------------------------------------------------------------------------------------
# Attention
# ------------------------------------------------------------------------------------------------
def _use_fused_attention(dtype, dk):
    """bf16 with 64-wide heads runs the fused kernels; fp32 (parity mode) and other head sizes the GEMM-composed path.
    S2T_ATTN_COMPOSED=1 forces the composed path (A/B measurements)."""
    return dtype == torch.bfloat16 and dk == 64 and os.environ.get("S2T_ATTN_COMPOSED", "0") != "1"


class AttentionFn(torch.autograd.Function):
    """out = residual + out_proj(softmax(scores) V) for
         kind == "abs": fairseq MultiheadAttention (modules/multihead_attention.py:161-431), self (fused q/k/v
                        projection) or encoder-decoder (q from xq, k/v from xkv), optional causal mask;
         kind == "rel": ESPnet RelPositionMultiHeadedAttention (modules/espnet_multihead_attention.py:313-356).

    The score matrix is materialised in fp32 ([B*H, Tq, Tk]) by a batched MFMA GEMM, turned into probabilities by
    the softmax kernel (mask / rel-shift / clamp fused) and consumed by a second batched GEMM.
    """

    @staticmethod
    def forward(ctx, xq, xkv, residual, prm, H, B, Tq, Tk, key_lens, causal, kind, pos_tab, train, drop_a, drop_o,
                ln_g=None, ln_b=None, pos_p=None, kv_all=None, kv_slot=None, q_rows=None):
        d = xq.shape[1]
        dk = d // H
        dt = xq.dtype
        dev = xq.device
        self_attn = xkv is None and kv_all is None
        # packed batches (s2t_amd/rows.py): ``key_lens`` names the geometry of the key side — and of the query side in
        # self-attention; ``q_rows`` that of the queries of an encoder-decoder attention (None: uniform rows)
        kr = key_lens if K.rows_geom(key_lens) is not None else None
        qr = kr if self_attn else (q_rows if K.rows_geom(q_rows) is not None else None)
        ctx.rows = (qr, kr)
        if (qr is not None or kr is not None) and not _use_fused_attention(dt, dk):
            raise NotImplementedError("packed rows: attention runs on the fused kernels only (bf16, head width 64)")
        ctx.kv_slot = kv_slot if kv_all is not None else None
        Mq, Mk = B * Tq, B * Tk
        ctx.ln = None
        if self_attn:
            wqkv = fused([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
            bqkv = fused_master([prm["q_b"], prm["k_b"], prm["v_b"]], 3 * d)
            qkv = torch.empty(Mq, 3 * d, dtype=dt, device=dev)
            if ln_g is not None:
                # xq is the block input BEFORE its LayerNorm (attention() only routes the row-block case here): the
                # LayerNorm rides in the projection kernel's prologue; the residual branch is xq itself
                x_pre, residual = xq, xq
                xq = torch.empty_like(x_pre) if train else None
                ln_mean = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                ln_rstd = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                K.rowblock_gemm(x_pre, wqkv, qkv, N=3 * d, ldc=3 * d, bias=bqkv, ln=(ln_g.data, ln_b.data), x_ln=xq,
                                ln_stats=(ln_mean, ln_rstd) if train else None, rows=qr)
                ctx.ln = (ln_g, ln_b, getattr(x_pre, "_s2t_drop_o", None))
                ctx.ln_saved = (x_pre, ln_mean, ln_rstd)
                if xq is None:
                    xq = x_pre  # eval: only shapes / dtypes are read below
            else:
                K.gemm(xq, wqkv, qkv, M=Mq, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d, bias=bqkv, rows=qr)
            q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
            ldq = ldk = 3 * d
        else:
            q = torch.empty(Mq, d, dtype=dt, device=dev)
            if ln_g is not None:
                # encoder-decoder attention behind its LayerNorm (attention() only routes the row-block case here): xq is the
                # block input, the LayerNorm rides in the query projection's prologue, the residual branch is xq itself
                x_pre, residual = xq, xq
                xq = torch.empty_like(x_pre) if train else None
                ln_mean = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                ln_rstd = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                K.rowblock_gemm(x_pre, cw(prm["q_w"]), q, N=d, ldc=d, bias=prm["q_b"].data, ln=(ln_g.data, ln_b.data), x_ln=xq,
                                ln_stats=(ln_mean, ln_rstd) if train else None, rows=qr)
                ctx.ln = (ln_g, ln_b, getattr(x_pre, "_s2t_drop_o", None))
                ctx.ln_saved = (x_pre, ln_mean, ln_rstd)
                if xq is None:
                    xq = x_pre
            else:
                K.gemm(xq, cw(prm["q_w"]), q, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, bias=prm["q_b"].data, rows=qr)
            if kv_all is not None:  # k | v of every decoder layer were projected by one launch (CrossKVFn): columns of layer l
                l, L, _ = kv_slot
                assert _use_fused_attention(dt, dk)
                k, v = kv_all[:, 2 * d * l:], kv_all[:, 2 * d * l + d:]
                ldq, ldk = d, 2 * d * L
            else:
                wkv = fused([prm["k_w"], prm["v_w"]], 2 * d, d)
                bkv = fused_master([prm["k_b"], prm["v_b"]], 2 * d)
                kv = torch.empty(Mk, 2 * d, dtype=dt, device=dev)
                K.gemm(xkv, wkv, kv, M=Mk, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, bias=bkv, rows=kr)
                k, v = kv, kv[:, d:]
                ldq, ldk = d, 2 * d
        Z = B * H
        if _use_fused_attention(dt, dk):
            # scores never leave the chip (csrc/attention_fused.hip)
            p = None
            if kind == "rel":
                assert self_attn and Tq == Tk
                n_pos = 2 * Tq - 1
                if pos_p is not None:  # projected for every layer of the stack by ONE batched launch (project_positions)
                    p = pos_p
                else:
                    p = torch.empty(n_pos, d, dtype=dt, device=dev)
                    K.gemm(pos_tab, cw(prm["pos_w"]), p, M=n_pos, N=d, K=d, lda=d, ldb=d, ldc=d)
                scale = 1.0 / math.sqrt(dk)
            else:
                scale = dk ** -0.5
            O = torch.empty(Mq, d, dtype=dt, device=dev)
            lse = torch.empty(Z, Tq, dtype=torch.float32, device=dev)
            # encoder-decoder attention in training: the rounding remainder of O rides along for the backward's delta (its score
            # gradient cancels over a few hundred nearly uniform keys; csrc/attention_fused.hip, attn_bwd_dq_kernel)
            O_lo = torch.empty(Mq, d, dtype=dt, device=dev) if (train and not self_attn and _ATTN_O_LO) else None
            K.attn_fused_fwd(q, Tq * ldq, ldq, k, Tk * ldk, ldk, v, Tk * ldk, ldk, O, Tq * d, d, lse, B, H, Tq, Tk, dk,
                             key_lens, causal, scale, p, d, prm["pos_u"].data.view(-1) if p is not None else None,
                             prm["pos_v"].data.view(-1) if p is not None else None, drop_a, q_rows=qr, k_rows=kr, o_lo=O_lo)
            ctx.o_lo = O_lo
            y = torch.empty(Mq, d, dtype=dt, device=dev)
            if _rb_ok(O, d) and (residual is None or (residual.stride(0) % 8 == 0 and residual.stride(1) == 1)):
                first = dict(x=O, w=cw(prm["o_w"]), out=y, N=d, ldc=d, bias=prm["o_b"].data, residual=residual,
                             ldr=residual.stride(0) if residual is not None else 0, drop=drop_o, rows=qr)
                req = _CHAIN["req"]
                if req is not None and self_attn and _rb_ok(y, 2 * d, "glu"):
                    cz = torch.empty(Mq, 2 * d, dtype=dt, device=dev) if train else None
                    cg = torch.empty(Mq, d, dtype=dt, device=dev)
                    cx = torch.empty(Mq, d, dtype=dt, device=dev) if train else None
                    cm = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                    cr = torch.empty(Mq, dtype=torch.float32, device=dev) if train else None
                    K.rowblock_chain(first, dict(x=y, w=req["w1"], out=cg, N=2 * d, ldc=d, act="glu", preact=cz, ldp=2 * d,
                                                 ln=(req["ln_g"].data, req["ln_b"].data), ln_lens=req["lens"], ln_T=req["T"],
                                                 x_ln=cx, ln_stats=(cm, cr) if train else None))
                    _CHAIN["out"] = {"key": (req["w1"].data_ptr(), req["ln_g"].data_ptr(), bool(train), id(req["lens"]), req["T"]),
                                     "g": cg, "z": cz, "x": cx, "mean": cm, "rstd": cr}
                else:
                    K.rowblock_gemm(first.pop("x"), first.pop("w"), first.pop("out"), **first)
            else:
                K.gemm(O, cw(prm["o_w"]), y, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, bias=prm["o_b"].data, residual=residual,
                       ldr=d, drop=drop_o, rows=qr)
            ctx.drops = (drop_a, drop_o)
            ctx.fused = True
            ctx.pos_pt = getattr(pos_p, "_s2t_pt", None) if (kind == "rel" and pos_p is not None) else None
            if train:
                ctx.save_for_backward(xq, xkv, q, k, v, O, lse, p, pos_tab, key_lens)
            ctx.prm, ctx.dims = prm, (H, B, Tq, Tk, d, dk, ldq, ldk, 0, 0, scale)
            ctx.kind, ctx.self_attn, ctx.has_res, ctx.causal = kind, self_attn, residual is not None, causal
            return y
        ctx.fused = False
        ldS = _pad8(Tk)
        S = torch.empty(Z, Tq, ldS, dtype=torch.float32, device=dev)
        BD = None
        ldB = 0
        qu = qv = p = None
        if kind == "rel":
            assert self_attn and Tq == Tk
            n_pos = 2 * Tq - 1
            ldB = _pad8(n_pos)
            p = torch.empty(n_pos, d, dtype=dt, device=dev)
            K.gemm(pos_tab, cw(prm["pos_w"]), p, M=n_pos, N=d, K=d, lda=d, ldb=d, ldc=d)
            qu = torch.empty(Mq, d, dtype=dt, device=dev)
            qv = torch.empty(Mq, d, dtype=dt, device=dev)
            K.bias_add_rows(q, ldq, prm["pos_u"].data, qu, d, Mq, d)
            K.bias_add_rows(q, ldq, prm["pos_v"].data, qv, d, Mq, d)
            K.gemm(qu, k, S, M=Tq, N=Tk, K=dk, lda=d, ldb=ldk, ldc=ldS, batch=Z, zdiv=H, a_s=(Tq * d, dk),
                   b_s=(Tk * ldk, dk), c_s=(H * Tq * ldS, Tq * ldS))
            BD = torch.empty(Z, Tq, ldB, dtype=torch.float32, device=dev)
            K.gemm(qv, p, BD, M=Tq, N=n_pos, K=dk, lda=d, ldb=d, ldc=ldB, batch=Z, zdiv=H, a_s=(Tq * d, dk),
                   b_s=(0, dk), c_s=(H * Tq * ldB, Tq * ldB))
            scale = 1.0 / math.sqrt(dk)
        else:
            K.gemm(q, k, S, M=Tq, N=Tk, K=dk, lda=ldq, ldb=ldk, ldc=ldS, batch=Z, zdiv=H, a_s=(Tq * ldq, dk),
                   b_s=(Tk * ldk, dk), c_s=(H * Tq * ldS, Tq * ldS))
            scale = dk ** -0.5
        P = torch.empty(Z, Tq, ldS, dtype=dt, device=dev)
        Pd = torch.empty(Z, Tq, ldS, dtype=dt, device=dev) if drop_a is not None else None
        K.attn_softmax_fwd(S, ldS, BD, ldB, P, ldS, Z, H, Tq, Tk, scale, key_lens, causal, kind == "rel", Pd, drop_a)
        del S, BD
        Pv = Pd if Pd is not None else P  # probabilities that multiply V (dropout applied)
        O = torch.empty(Mq, d, dtype=dt, device=dev)
        K.gemm(Pv, v, O, M=Tq, N=dk, K=Tk, lda=ldS, ldb=ldk, ldc=d, b_kmajor=True, batch=Z, zdiv=H,
               a_s=(H * Tq * ldS, Tq * ldS), b_s=(Tk * ldk, dk), c_s=(Tq * d, dk))
        y = torch.empty(Mq, d, dtype=dt, device=dev)
        K.gemm(O, cw(prm["o_w"]), y, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, bias=prm["o_b"].data, residual=residual, ldr=d,
               drop=drop_o)
        ctx.drops = (drop_a, drop_o)
        if train:
            ctx.save_for_backward(xq, xkv, q, k, v, P, O, qu, qv, p, pos_tab, Pd)
        ctx.prm, ctx.dims = prm, (H, B, Tq, Tk, d, dk, ldq, ldk, ldS, ldB, scale)
        ctx.kind, ctx.self_attn, ctx.has_res = kind, self_attn, residual is not None
        return y

    @staticmethod
    def _backward_fused(ctx, dy):
        xq, xkv, q, k, v, O, lse, p, pos_tab, key_lens = ctx.saved_tensors
        prm = ctx.prm
        H, B, Tq, Tk, d, dk, ldq, ldk, _, _, scale = ctx.dims
        dt, dev = xq.dtype, xq.device
        Mq, Mk = B * Tq, B * Tk
        Z = B * H
        drop_a, drop_o = ctx.drops
        dres = dy.contiguous()
        dy = _drop_rows(dres, drop_o)
        qr, kr = ctx.rows
        dO = _dgrad_rowblock(dy, prm["o_w"], d, rows=qr) if d == 256 else None
        if dO is None:
            dO = torch.empty(Mq, d, dtype=dt, device=dev)
            K.gemm(dy, cw(prm["o_w"]), dO, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, b_kmajor=True, rows=qr)
        _wgrad(dy, O, prm["o_w"].grad, d, d, Mq, d, d, 1.0, prm["o_b"].grad, rows=qr)
        _ready(prm["o_w"], prm["o_b"])
        if ctx.self_attn:
            dqkv = torch.empty(Mq, 3 * d, dtype=dt, device=dev)
            dq, dk_, dv = dqkv, dqkv[:, d:], dqkv[:, 2 * d:]
        else:
            dq = torch.empty(Mq, d, dtype=dt, device=dev)
            if ctx.kv_slot is not None:  # this layer's columns of the stack's shared [Mk, L*2d] gradient (CrossKVFn)
                l, L, share = ctx.kv_slot
                if share.get("dkv") is None:
                    share["dkv"], share["done"] = torch.empty(Mk, 2 * d * L, dtype=dt, device=dev), 0
                dkv = share["dkv"]
                dk_, dv = dkv[:, 2 * d * l:], dkv[:, 2 * d * l + d:]
            else:
                dkv = torch.empty(Mk, 2 * d, dtype=dt, device=dev)
                dk_, dv = dkv, dkv[:, d:]
        rel = ctx.kind == "rel"
        pt = getattr(ctx, "pos_pt", None) if (rel and _ATTN_DQV_FUSED) else None
        one_pass = False
        if (rel and _RELPOS_ONE_PASS and _RELPOS_GLUE and pt is None and dt == torch.bfloat16 and dk == 64 and Tq == Tk
                and Tq <= 256 and not ctx.causal and ldq % 8 == 0 and _arm_backward_end()):
            # sequences of up to 256 frames: scores, position band and exponentials formed ONCE, the skewed score gradient kept
            # on the chip (csrc/relpos_bwd.hip) — dq complete, dk, dv, both bias-gradient column sums and this layer's
            # per-utterance tables of the position gradient from one launch (no dbd / delta / Q + pos_bias_v buffers)
            n_pos = 2 * Tq - 1
            ws = _ln_workspace(d, dev)
            dp = _posq_slot(n_pos, d, dev, zero=False)
            slot = len(_POSQ["parts"]) if (_FOLD_DEFER and len(_POSQ["parts"]) < _FOLD_CAP) else None
            part = K.relpos_attn_bwd(q, Tq * ldq, ldq, k, Tk * ldk, ldk, v, Tk * ldk, ldk, O, dO, Tq * d, d, lse, dq, dk_, dv, p,
                                     d, prm["pos_u"].data.view(-1), prm["pos_v"].data.view(-1), ws, ws[d:], B, H, Tq, dk, key_lens,
                                     scale, drop_a, replicas=K.LN_REPLICAS, replica_stride=2 * d, defer_slot=slot, rows=qr)
            if slot is not None:
                _POSQ["parts"].append((part, dp, (B, H, Tq, dk)))
            else:
                K.relpos_dp_reduce([part], [dp], B, H, Tq, dk)
            _LNQ["entries"].append((ws, prm["pos_u"].grad.view(-1), prm["pos_v"].grad.view(-1), d))
            _POSQ["entries"].append((dp, _pos_table_f32(pos_tab), prm["pos_w"].grad, n_pos, d))
            _ready(prm["pos_w"], prm["pos_u"], prm["pos_v"])
            one_pass = True
        if (not rel and _ATTN_ONE_PASS and dt == torch.bfloat16 and dk == 64 and _ATTN_ONE_PASS_MIN_TK <= Tk <= 256 and ldq % 8 == 0
                and ldk % 8 == 0):
            # plain attention over 160 - 256 keys (encoder-decoder attention, the plain Transformer encoder): dq, dk, dv from one
            # launch on the schedule of the relative-position kernel (csrc/relpos_bwd.hip without its position terms).  Fewer keys
            # leave most of its eight key-owning waves idle: the decoder's self-attention (61 x 61: 13.8 us on the two kernels,
            # 14.6 in one pass; 250 x 250: 52.6 / 42.7; 61 x 250: 23.4 / 17.5 — tools/attn_bwd_routes.py) stays on two kernels
            K.attn_bwd_one_pass(q, Tq * ldq, ldq, k, Tk * ldk, ldk, v, Tk * ldk, ldk, O, dO, Tq * d, d, lse, dq, dk_, dv, B, H, Tq, Tk,
                                dk, key_lens, ctx.causal, scale, drop_a, q_rows=qr, k_rows=kr, o_lo=getattr(ctx, "o_lo", None))
            ctx.o_lo = None
            one_pass = True
        if not one_pass:
            delta = torch.empty(Z, Tq, dtype=torch.float32, device=dev)
            n_pos = 2 * Tq - 1
            ldB = _pad8(n_pos)
            dBD = _dbd_buffer(H, B, Tq, ldB, dt, dev) if rel else None
            qv = torch.empty(Mq, d, dtype=dt, device=dev) if rel else None  # q + pos_bias_v, written by the dQ kernel
            K.attn_fused_bwd(q, Tq * ldq, ldq, k, Tk * ldk, ldk, v, Tk * ldk, ldk, O, dO, Tq * d, d, lse, delta, dq, dk_, dv, dBD,
                             ldB, B, H, Tq, Tk, dk, key_lens, ctx.causal, scale, p, d,
                             prm["pos_u"].data.view(-1) if rel else None, prm["pos_v"].data.view(-1) if rel else None, drop_a,
                             dbd_band_only=rel, pos_pt=pt[0] if pt is not None else None, pt_ld=pt[1] if pt is not None else 0,
                             dpos_u=prm["pos_u"].grad.view(-1) if pt is not None else None,
                             dpos_v=prm["pos_v"].grad.view(-1) if pt is not None else None, qv_out=qv, q_rows=qr, k_rows=kr,
                             o_lo=getattr(ctx, "o_lo", None))
            ctx.o_lo = None
            glue_done = False
            if rel and _RELPOS_GLUE and pt is None and dt == torch.bfloat16 and dk == 64 and Tq <= _GLUE_MAX_T and ldq % 4 == 0 and _arm_backward_end():
                # everything behind dbd in ONE pass over it (csrc/relpos_glue.hip): the (Q+v) branch added into dq, both bias
                # gradients (column sums into the replicated workspace, folded with the LayerNorm gradients) and this layer's
                # gradient w.r.t. the projected positions, queued for the batched linear_pos weight gradient
                ws = _ln_workspace(d, dev)
                dp = _posq_slot(n_pos, d, dev, zero=False)
                slot = len(_POSQ["parts"]) if (_FOLD_DEFER and len(_POSQ["parts"]) < _FOLD_CAP) else None
                part = K.relpos_glue(dBD, ldB, p, d, qv, dq, Tq * ldq, ldq, ws, ws[d:], dp, B, H, Tq, dk, replicas=K.LN_REPLICAS,
                                     replica_stride=2 * d, defer_slot=slot, rows=qr)
                if slot is not None:
                    _POSQ["parts"].append((part, dp, (B, H, Tq, dk)))
                _LNQ["entries"].append((ws, prm["pos_u"].grad.view(-1), prm["pos_v"].grad.view(-1), d))
                _POSQ["entries"].append((dp, _pos_table_f32(pos_tab), prm["pos_w"].grad, n_pos, d))
                _ready(prm["pos_w"], prm["pos_u"], prm["pos_v"])
                glue_done = True
            if rel and not glue_done and qr is not None:
                raise NotImplementedError("packed rows: the relative-position backward runs through s2t_relpos_glue only")
            if rel and not glue_done:
                fuse_glue = dt == torch.bfloat16 and d == 256 and ldq % 8 == 0
                # the (Q+v) branch, the add into dq and both bias gradients in one band-limited launch (s2t_relpos_dqv) when the
                # stack kept the transposed projections
                pt_glue = getattr(ctx, "pos_pt", None) if (pt is None and _RELPOS_DQV and dt == torch.bfloat16 and ldq % 4 == 0) else None
                if pt_glue is not None:
                    if _arm_backward_end():  # column sums into the replicated workspace, folded with the LayerNorm gradients
                        ws = _ln_workspace(d, dev)
                        K.relpos_dqv(dBD, ldB, pt_glue[0], pt_glue[1], dq, Tq * ldq, ldq, ws, ws[d:], B, H, Tq, dk,
                                     replicas=K.LN_REPLICAS, replica_stride=2 * d)
                        _LNQ["entries"].append((ws, prm["pos_u"].grad.view(-1), prm["pos_v"].grad.view(-1), d))
                    else:
                        K.relpos_dqv(dBD, ldB, pt_glue[0], pt_glue[1], dq, Tq * ldq, ldq, prm["pos_u"].grad.view(-1),
                                     prm["pos_v"].grad.view(-1), B, H, Tq, dk)
                elif pt is None:
                    if not fuse_glue:
                        K.colsum_accum(dq, ldq, prm["pos_u"].grad.view(-1), Mq, d)
                    dqv = torch.empty(Mq, d, dtype=dt, device=dev)
                    K.gemm(dBD, p, dqv, M=Tq, N=dk, K=n_pos, lda=ldB, ldb=d, ldc=d, b_kmajor=True, batch=Z, zdiv=H,
                           a_s=(Tq * ldB, B * Tq * ldB), b_s=(0, dk), c_s=(Tq * d, dk))
                    if not fuse_glue:
                        K.colsum_accum(dqv, d, prm["pos_v"].grad.view(-1), Mq, d)
                ktiles = (Mq + 63) // 64
                sk = max(1, min(ktiles, _DP_SPLIT))
                # two-phase split-K in overwrite mode (c_atomic = 2) needs no zero fill of dp
                pos32 = _pos_table_f32(pos_tab)
                queued = _arm_backward_end()
                dp = _posq_slot(n_pos, d, dev, zero=sk <= 1) if queued else \
                    (torch.empty if sk > 1 else torch.zeros)(n_pos, d, dtype=torch.float32, device=dev)
                K.gemm(dBD, qv, dp, M=n_pos, N=dk, K=Mq, lda=ldB, ldb=d, ldc=d, a_kmajor=True, b_kmajor=True, batch=H, zdiv=1,
                       a_s=(B * Tq * ldB, 0), b_s=(dk, 0), c_s=(dk, 0), split_k=sk, c_atomic=2 if sk > 1 else True)
                if queued:  # linear_pos weight gradients of all layers: one batched launch at the end of the pass
                    _POSQ["entries"].append((dp, pos32, prm["pos_w"].grad, n_pos, d))
                else:
                    K.gemm(dp, pos32, prm["pos_w"].grad, M=d, N=d, K=n_pos, lda=d, ldb=d, ldc=d, a_kmajor=True, b_kmajor=True,
                           split_k=_POSW_SPLIT if _POSW_SPLIT else max(1, min(8, (n_pos + 63) // 64)), c_atomic=True)
                if pt is not None or pt_glue is not None:
                    pass  # the kernel wrote the complete dq and added both bias gradients
                elif fuse_glue:  # dq += dqv, pos_u.grad += colsum(dq), pos_v.grad += colsum(dqv) in one pass
                    K.add_colsum2(dq, ldq, dqv, d, prm["pos_u"].grad.view(-1), prm["pos_v"].grad.view(-1), Mq, d)
                else:
                    dq[:, :d].add_(dqv)
                _ready(prm["pos_w"], prm["pos_u"], prm["pos_v"])
        dxq = None
        dx_ln = None
        if ctx.self_attn:
            gw = fused_grad([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
            gb = prm["q_b"].grad.as_strided((3 * d,), (1,))
            wqkv = fused([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
            _wgrad(dqkv, xq, gw, 3 * d, d, Mq, 3 * d, d, 1.0, gb, rows=qr)
            if ctx.ln is not None:  # projection dgrad + the LayerNorm's backward in one row-block launch
                ln_g, ln_b, up_drop = ctx.ln
                x_pre, ln_mean, ln_rstd = ctx.ln_saved
                dx_ln = _dgrad_ln_backward(dqkv, prm["q_w"], 3 * d, x_pre, ln_g, ln_b, ln_mean, ln_rstd, None, 0, dres, up_drop,
                                           rows=qr)
            if dx_ln is None:
                dxq = torch.empty(Mq, d, dtype=dt, device=dev)
                K.gemm(dqkv, wqkv, dxq, M=Mq, N=d, K=3 * d, lda=3 * d, ldb=d, ldc=d, b_kmajor=True, rows=qr)
            dxkv = None
        else:
            _wgrad(dq, xq, prm["q_w"].grad, d, d, Mq, d, d, 1.0, prm["q_b"].grad, rows=qr)
            if ctx.ln is not None:  # query-projection dgrad + the LayerNorm's backward in one row-block launch
                ln_g, ln_b, up_drop = ctx.ln
                x_pre, ln_mean, ln_rstd = ctx.ln_saved
                dx_ln = _dgrad_ln_backward(dq, prm["q_w"], d, x_pre, ln_g, ln_b, ln_mean, ln_rstd, None, 0, dres, up_drop, rows=qr)
            dxq = _dgrad_rowblock(dq, prm["q_w"], d, rows=qr) if (d == 256 and dx_ln is None) else None
            if dxq is None and dx_ln is None:
                dxq = torch.empty(Mq, d, dtype=dt, device=dev)
                K.gemm(dq, cw(prm["q_w"]), dxq, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, b_kmajor=True, rows=qr)
            if ctx.kv_slot is not None:
                # weight and memory gradients of the k | v projections: CrossKVFn.backward, once for the stack.  The layer
                # that completes the shared buffer hands it on as the gradient of kv_all; the others contribute nothing
                # autograd would have to add.
                l, L, share = ctx.kv_slot
                share["done"] += 1
                dkv_all = None
                if share["done"] == L:
                    dkv_all, share["dkv"], share["done"] = share["dkv"], None, 0
                _ready(prm["q_w"], prm["q_b"])
                if ctx.ln is not None:
                    if dx_ln is None:
                        ln_g, ln_b, up_drop = ctx.ln
                        x_pre, ln_mean, ln_rstd = ctx.ln_saved
                        dx_ln = _ln_backward(x_pre, ln_g, ln_b, dxq, ln_mean, ln_rstd, None, 0, dres, up_drop, rows=qr)
                    return (dx_ln, None, None) + (None,) * 15 + (dkv_all, None, None)
                return (dxq, None, (dres if ctx.has_res else None)) + (None,) * 15 + (dkv_all, None, None)
            gw = fused_grad([prm["k_w"], prm["v_w"]], 2 * d, d)
            gb = prm["k_b"].grad.as_strided((2 * d,), (1,))
            wkv = fused([prm["k_w"], prm["v_w"]], 2 * d, d)
            _wgrad(dkv, xkv, gw, 2 * d, d, Mk, 2 * d, d, 1.0, gb, rows=kr)
            dxkv = torch.empty(Mk, d, dtype=dt, device=dev)
            K.gemm(dkv, wkv, dxkv, M=Mk, N=d, K=2 * d, lda=2 * d, ldb=d, ldc=d, b_kmajor=True, rows=kr)
        _ready(prm["q_w"], prm["k_w"], prm["v_w"], prm["q_b"], prm["k_b"], prm["v_b"])
        if dx_ln is not None:
            return (dx_ln, None, None) + (None,) * 18
        if ctx.ln is not None:
            ln_g, ln_b, up_drop = ctx.ln
            x_pre, ln_mean, ln_rstd = ctx.ln_saved
            dx = _ln_backward(x_pre, ln_g, ln_b, dxq, ln_mean, ln_rstd, None, 0, dres, up_drop, rows=qr)
            return (dx, None, None) + (None,) * 18
        return (dxq, dxkv, (dres if ctx.has_res else None)) + (None,) * 18

    @staticmethod
    def backward(ctx, dy):
        if ctx.fused:
            return AttentionFn._backward_fused(ctx, dy)
        xq, xkv, q, k, v, P, O, qu, qv, p, pos_tab, Pd = ctx.saved_tensors
        prm = ctx.prm
        H, B, Tq, Tk, d, dk, ldq, ldk, ldS, ldB, scale = ctx.dims
        dt, dev = xq.dtype, xq.device
        Mq, Mk = B * Tq, B * Tk
        Z = B * H
        drop_a, drop_o = ctx.drops
        dres = dy.contiguous()
        dy = _drop_rows(dres, drop_o)
        Pv = Pd if Pd is not None else P
        # out_proj
        dO = torch.empty(Mq, d, dtype=dt, device=dev)
        K.gemm(dy, cw(prm["o_w"]), dO, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, b_kmajor=True)
        _wgrad(dy, O, prm["o_w"].grad, d, d, Mq, d, d, 1.0, prm["o_b"].grad)
        _ready(prm["o_w"], prm["o_b"])
        # gradient buffers in the projection layout so that one GEMM handles dW / dx
        if ctx.self_attn:
            dqkv = torch.empty(Mq, 3 * d, dtype=dt, device=dev)
            dq, dk_, dv = dqkv, dqkv[:, d:], dqkv[:, 2 * d:]
        else:
            dq = torch.empty(Mq, d, dtype=dt, device=dev)
            dkv = torch.empty(Mk, 2 * d, dtype=dt, device=dev)
            dk_, dv = dkv, dkv[:, d:]
        # dP = dO V^T ; dV = P^T dO
        dP = torch.empty(Z, Tq, ldS, dtype=torch.float32, device=dev)
        K.gemm(dO, v, dP, M=Tq, N=Tk, K=dk, lda=d, ldb=ldk, ldc=ldS, batch=Z, zdiv=H, a_s=(Tq * d, dk),
               b_s=(Tk * ldk, dk), c_s=(H * Tq * ldS, Tq * ldS))
        K.gemm(Pv, dO, dv, M=Tk, N=dk, K=Tq, lda=ldS, ldb=d, ldc=ldk, a_kmajor=True, b_kmajor=True, batch=Z, zdiv=H,
               a_s=(H * Tq * ldS, Tq * ldS), b_s=(Tq * d, dk), c_s=(Tk * ldk, dk))
        dS = torch.empty(Z, Tq, ldS, dtype=dt, device=dev)
        dBD = torch.empty(H, B, Tq, ldB, dtype=dt, device=dev) if ctx.kind == "rel" else None
        K.attn_softmax_bwd(P, ldS, dP, ldS, dS, ldS, dBD, ldB, Z, H, Tq, Tk, scale, drop_a)
        del dP
        qa = qu if ctx.kind == "rel" else q  # the matrix that multiplied K^T in the forward
        lda_q = d if ctx.kind == "rel" else ldq
        # dQ(ac) = dS K ; dK = dS^T (q [+u])
        K.gemm(dS, k, dq, M=Tq, N=dk, K=Tk, lda=ldS, ldb=ldk, ldc=ldq, b_kmajor=True, batch=Z, zdiv=H,
               a_s=(H * Tq * ldS, Tq * ldS), b_s=(Tk * ldk, dk), c_s=(Tq * ldq, dk))
        K.gemm(dS, qa, dk_, M=Tk, N=dk, K=Tq, lda=ldS, ldb=lda_q, ldc=ldk, a_kmajor=True, b_kmajor=True, batch=Z, zdiv=H,
               a_s=(H * Tq * ldS, Tq * ldS), b_s=(Tq * lda_q, dk), c_s=(Tk * ldk, dk))
        if ctx.kind == "rel":
            n_pos = 2 * Tq - 1
            # pos_bias_u gradient = column sums of d(q+u) (only the ac term so far)
            K.colsum_accum(dq, ldq, prm["pos_u"].grad.view(-1), Mq, d)
            # d(q+v) = dBD p  (per head); kept separately for pos_bias_v, then added into dq
            dqv = torch.empty(Mq, d, dtype=dt, device=dev)
            K.gemm(dBD, p, dqv, M=Tq, N=dk, K=n_pos, lda=ldB, ldb=d, ldc=d, b_kmajor=True, batch=Z, zdiv=H,
                   a_s=(Tq * ldB, B * Tq * ldB), b_s=(0, dk), c_s=(Tq * d, dk))
            K.colsum_accum(dqv, d, prm["pos_v"].grad.view(-1), Mq, d)
            # dp[n, h, :] = sum_{b,i} dBD[h, (b,i), n] * (q+v)[(b,i), h, :]   (one GEMM per head, K = B*Tq)
            ktiles = (Mq + 63) // 64
            sk = max(1, min(ktiles, _DP_SPLIT))
            # two-phase split-K in overwrite mode (c_atomic = 2) needs no zero fill of dp
            dp = (torch.empty if sk > 1 else torch.zeros)(n_pos, d, dtype=torch.float32, device=dev)
            K.gemm(dBD, qv, dp, M=n_pos, N=dk, K=Mq, lda=ldB, ldb=d, ldc=d, a_kmajor=True, b_kmajor=True, batch=H, zdiv=1,
                   a_s=(B * Tq * ldB, 0), b_s=(dk, 0), c_s=(dk, 0), split_k=sk, c_atomic=2 if sk > 1 else True)
            # linear_pos weight: dW[dout, din] += dp^T pos_tab  (fp32 GEMM on the fp32 table)
            pos32 = _pos_table_f32(pos_tab)
            K.gemm(dp, pos32, prm["pos_w"].grad, M=d, N=d, K=n_pos, lda=d, ldb=d, ldc=d, a_kmajor=True, b_kmajor=True,
                   split_k=max(1, min(8, (n_pos + 63) // 64)), c_atomic=True)
            # dq += dqv   (strided add into the q slice of dqkv)
            dq_view = dq[:, :d] if ctx.self_attn else dq
            dq_view.add_(dqv)
            _ready(prm["pos_w"], prm["pos_u"], prm["pos_v"])
        # projections
        dxq = torch.empty(Mq, d, dtype=dt, device=dev)
        if ctx.self_attn:
            gw = fused_grad([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
            gb = prm["q_b"].grad.as_strided((3 * d,), (1,))
            wqkv = fused([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
            _wgrad(dqkv, xq, gw, 3 * d, d, Mq, 3 * d, d, 1.0, gb)
            K.gemm(dqkv, wqkv, dxq, M=Mq, N=d, K=3 * d, lda=3 * d, ldb=d, ldc=d, b_kmajor=True)
            dxkv = None
        else:
            _wgrad(dq, xq, prm["q_w"].grad, d, d, Mq, d, d, 1.0, prm["q_b"].grad)
            K.gemm(dq, cw(prm["q_w"]), dxq, M=Mq, N=d, K=d, lda=d, ldb=d, ldc=d, b_kmajor=True)
            gw = fused_grad([prm["k_w"], prm["v_w"]], 2 * d, d)
            gb = prm["k_b"].grad.as_strided((2 * d,), (1,))
            wkv = fused([prm["k_w"], prm["v_w"]], 2 * d, d)
            _wgrad(dkv, xkv, gw, 2 * d, d, Mk, 2 * d, d, 1.0, gb)
            dxkv = torch.empty(Mk, d, dtype=dt, device=dev)
            K.gemm(dkv, wkv, dxkv, M=Mk, N=d, K=2 * d, lda=2 * d, ldb=d, ldc=d, b_kmajor=True)
        _ready(prm["q_w"], prm["k_w"], prm["v_w"], prm["q_b"], prm["k_b"], prm["v_b"])
        return (dxq, dxkv, (dres if ctx.has_res else None)) + (None,) * 18


_PT = {}
_PT_OFF = 16  # zero columns in front of position 0 of the transposed projections (windows start up to 15 positions early)
# (Q+v) branch of dQ and the two bias gradients inside the attention backward kernel (s2t_attn_fused_bwd, pos_pt).  Opt-in:
# measured on the headline shape the dQ kernel grows from 54 to 119 us (12 more MFMAs per key block, but 250 registers and
# a longer dependent chain in a kernel that is latency-bound already), more than the 46 us of the GEMM + column-sum launches
# it replaces (profiles/README.md, r02 notes)
_ATTN_DQV_FUSED = os.environ.get("S2T_ATTN_DQV_FUSED", "0") == "1"
# the same branch as ONE launch behind the attention backward kernels (s2t_relpos_dqv: band-limited product from the dBD rows
# and the transposed projections, dq += dqv and both bias gradients in its epilogue) in place of a batched GEMM over the
# half-empty dBD rows and the add + column-sum pass
_RELPOS_DQV = os.environ.get("S2T_RELPOS_DQV", "1") != "0"


def project_positions(pos_tab, weights):
    """linear_pos (espnet_multihead_attention.py:331) of EVERY layer of a stack in one batched launch: the relative-position
    table is the same for all of them and the [d, d] weights sit at a constant stride in the flat parameter buffer.
    Returns one [2T-1, d] matrix per layer, or None when the layout (or dtype) does not allow it."""
    if not weights or pos_tab is None or pos_tab.dtype != torch.bfloat16 or not pos_tab.is_cuda:
        return None
    ws = [cw(w) for w in weights]
    d = ws[0].shape[1]
    if any(w.shape != (d, d) or w.dtype != torch.bfloat16 for w in ws):
        return None
    esz = ws[0].element_size()
    if len(ws) > 1:
        stride = (ws[1].data_ptr() - ws[0].data_ptr()) // esz
        if stride <= 0 or stride % 8 or any((ws[i].data_ptr() - ws[0].data_ptr()) != i * stride * esz for i in range(len(ws))):
            return None
    else:
        stride = 0
    n_pos = pos_tab.shape[0]
    out = torch.empty(len(ws), n_pos, d, dtype=torch.bfloat16, device=pos_tab.device)
    K.gemm(pos_tab, ws[0], out, M=n_pos, N=d, K=d, lda=d, ldb=d, ldc=d, batch=len(ws), a_s=(0, 0), b_s=(stride, 0),
           c_s=(n_pos * d, 0))
    res = [out[i] for i in range(len(ws))]
    if torch.is_grad_enabled() and (_ATTN_DQV_FUSED or _RELPOS_DQV):
        # the TRANSPOSED projections for the backward kernel (s2t_attn_fused_bwd, pos_pt): [layer][d][PT_OFF + n_pos + pad],
        # zero outside the n_pos valid columns (the buffer is kept per stack: the GEMM rewrites exactly the valid part)
        ld = _pad8(_PT_OFF + n_pos + 96)
        key = (str(pos_tab.device), ws[0].data_ptr(), len(ws), n_pos, d)
        buf = _PT.get(key)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                return res
            while len(_PT) >= 8:
                _PT.pop(next(iter(_PT)))
            buf = _PT[key] = torch.zeros(len(ws), d, ld, dtype=torch.bfloat16, device=pos_tab.device)
        K.gemm(ws[0], pos_tab, buf[0, :, _PT_OFF:], M=d, N=n_pos, K=d, lda=d, ldb=d, ldc=ld, batch=len(ws), a_s=(stride, 0),
               b_s=(0, 0), c_s=(d * ld, 0))
        for i, r in enumerate(res):
            r._s2t_pt = (buf[i, :, _PT_OFF:], ld)
    return res


_ATTN_O_LO = os.environ.get("S2T_ATTN_O_LO", "1") != "0"  # rounding remainder of the encoder-decoder attention output for delta
_CROSS_KV = os.environ.get("S2T_CROSS_KV", "1") != "0"  # one k | v projection of the encoder memory for the whole decoder


class CrossKVFn(torch.autograd.Function):
    """k | v projections of the encoder output for the encoder-decoder attention of EVERY decoder layer
    (modules/multihead_attention.py:205-215 with static key / value = encoder_out, once per layer in the reference):
    kv_all[:, 2d*l : 2d*(l+1)] = mem W_kv,l^T + b_kv,l.  One GEMM with N = L*2d in place of L launches; in the backward the
    L input gradients (and the L-1 additions autograd would make) are one GEMM with K = L*2d over the layers' shared
    gradient buffer, which the attention kernels fill column block by column block (AttentionFn, kv_slot)."""

    @staticmethod
    def forward(ctx, mem, prms, rows=None):
        Mk, d = mem.shape
        L = len(prms)
        # the layers' [2d, d] weight pairs sit apart in the flat buffer: gather them (L * 256 KiB) into one operand
        w_all = torch.cat([fused([p["k_w"], p["v_w"]], 2 * d, d) for p in prms], 0)
        b_all = torch.cat([fused_master([p["k_b"], p["v_b"]], 2 * d) for p in prms], 0)
        kv_all = torch.empty(Mk, 2 * d * L, dtype=mem.dtype, device=mem.device)
        K.gemm(mem, w_all, kv_all, M=Mk, N=2 * d * L, K=d, lda=d, ldb=d, ldc=2 * d * L, bias=b_all, rows=rows)
        ctx.save_for_backward(mem, w_all)
        ctx.prms, ctx.rows = prms, rows
        return kv_all

    @staticmethod
    def backward(ctx, dkv_all):
        mem, w_all = ctx.saved_tensors
        Mk, d = mem.shape
        L = len(ctx.prms)
        for l, prm in enumerate(ctx.prms):
            gw = fused_grad([prm["k_w"], prm["v_w"]], 2 * d, d)
            gb = prm["k_b"].grad.as_strided((2 * d,), (1,))
            _wgrad(dkv_all[:, 2 * d * l:], mem, gw, 2 * d, d, Mk, 2 * d * L, d, 1.0, gb, rows=ctx.rows)
            _ready(prm["k_w"], prm["v_w"], prm["k_b"], prm["v_b"])
        dmem = torch.empty(Mk, d, dtype=mem.dtype, device=mem.device)
        _dgrad(dkv_all, w_all, dmem, Mk, d, 2 * d * L, 2 * d * L, d, d, rows=ctx.rows)
        return dmem, None, None


def cross_kv(mem, prms, H, rows=None):
    """(kv_all, share) for ``attention(..., kv=(kv_all, l, L, share))`` of the L decoder layers, or None where the fused
    attention kernels (which take the strided k / v) do not apply."""
    d = mem.shape[1]
    if not (_CROSS_KV and mem.is_cuda and len(prms) > 1 and _use_fused_attention(mem.dtype, d // H) and mem.is_contiguous()):
        return None
    return CrossKVFn.apply(mem, prms, rows if K.rows_geom(rows) is not None else None), {}


_CHAIN = {"req": None, "out": None}


def attention(xq, xkv, residual, prm, H, B, Tq, Tk, key_lens=None, causal=False, kind="abs", pos_tab=None,
              p_attn=0.0, p_out=0.0, training=False, ln=None, pos_p=None, kv=None, q_rows=None, chain=None):
    """``ln`` = (gamma, beta) of the LayerNorm in front of a SELF-attention block: ``xq`` is then the block input before
    that LayerNorm and doubles as the residual (``residual`` must be None).  Where the row-block projection kernel applies
    the LayerNorm rides in its prologue; otherwise it runs as its own kernel first."""
    if ln is not None:
        # self-attention, or encoder-decoder attention on keys / values projected for the whole stack (``kv``)
        assert xkv is None and residual is None
        d = xq.shape[1]
        if not (_use_fused_attention(xq.dtype, d // H) and _rb_ok(xq, 3 * d)):
            y, xr = layer_norm(xq, ln[0], ln[1], fork=True, rows=q_rows if kv is not None else key_lens)
            return attention(y, None, xr, prm, H, B, Tq, Tk, key_lens, causal, kind, pos_tab, p_attn, p_out, training,
                             pos_p=pos_p, kv=kv, q_rows=q_rows)
    drop_a = DROPOUT.next(p_attn if training else 0.0, xq.device)
    drop_o = DROPOUT.next(p_out if training else 0.0, xq.device)
    lg, lb = ln if ln is not None else (None, None)
    if not (kind == "rel" and _use_fused_attention(xq.dtype, xq.shape[1] // H)):
        pos_p = None
    if kv is not None:  # pre-projected keys / values of the whole decoder stack (cross_kv)
        kv_all, l, L, share = kv
        return _tag_drop(AttentionFn.apply(xq, None, residual, prm, H, B, Tq, Tk, key_lens, causal, kind, pos_tab,
                                           torch.is_grad_enabled(), drop_a, drop_o, lg, lb, pos_p, kv_all, (l, L, share),
                                           q_rows), drop_o)
    # ``chain`` (a Conformer layer: dict(w1, ln_g, ln_b, lens, T) of the convolution module that follows): the output projection
    # and conv_norm + pointwise conv 1 + GLU run as ONE launch (s2t_rowblock_chain); what the module's first stage would have
    # produced rides on the output tensor, ConvModuleFn picks it up instead of launching
    _CHAIN["req"], _CHAIN["out"] = (chain if K.RB_CHAIN else None), None
    try:
        y = _tag_drop(AttentionFn.apply(xq, xkv, residual, prm, H, B, Tq, Tk, key_lens, causal, kind, pos_tab,
                                        torch.is_grad_enabled(), drop_a, drop_o, lg, lb, pos_p, None, None, q_rows), drop_o)
    finally:
        _CHAIN["req"] = None
    if _CHAIN["out"] is not None:
        y._s2t_chain, _CHAIN["out"] = _CHAIN["out"], None
    return y


# ------------------------------------------------------------------------------------------------
# Incremental decoding (beam search): one query position per sequence against cached keys / values
# ------------------------------------------------------------------------------------------------
def attention_step(xq, residual, prm, H, kv, n_keys, key_lens, self_attention):
    """One decoding step of fairseq MultiheadAttention with incremental state (modules/multihead_attention.py:302-339):
    ``xq`` [Bb, d] is the (layer-normed) current position of every hypothesis, ``kv`` [Bb, cap, 2d] the cache of
    projected keys | values.  Self-attention appends this position's k | v at row ``n_keys - 1`` (projected here, by
    the same fused QKV GEMM as training); encoder-decoder attention reads the static projected memory.  Inference only
    (no autograd).  Returns residual + out_proj(softmax(q k^T / sqrt(dk)) v)."""
    Bb, d = xq.shape
    dk = d // H
    dt, dev = xq.dtype, xq.device
    cap = kv.shape[1]
    if self_attention:
        wqkv = fused([prm["q_w"], prm["k_w"], prm["v_w"]], 3 * d, d)
        bqkv = fused_master([prm["q_b"], prm["k_b"], prm["v_b"]], 3 * d)
        qkv = torch.empty(Bb, 3 * d, dtype=dt, device=dev)
        K.gemm(xq, wqkv, qkv, M=Bb, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d, bias=bqkv)
        kv[:, n_keys - 1].copy_(qkv[:, d:])
        q, ldq = qkv, 3 * d
    else:
        q = torch.empty(Bb, d, dtype=dt, device=dev)
        K.gemm(xq, cw(prm["q_w"]), q, M=Bb, N=d, K=d, lda=d, ldb=d, ldc=d, bias=prm["q_b"].data)
        ldq = d
    k, v = kv, kv[:, :, d:]
    kv_sb, ldk = cap * 2 * d, 2 * d
    Z = Bb * H
    scale = dk ** -0.5
    O = torch.empty(Bb, d, dtype=dt, device=dev)
    if _use_fused_attention(dt, dk):
        K.attn_fused_fwd(q, ldq, ldq, k, kv_sb, ldk, v, kv_sb, ldk, O, d, d, None, Bb, H, 1, n_keys, dk, key_lens, False, scale)
    else:
        ldS = _pad8(n_keys)
        S = torch.empty(Z, 1, ldS, dtype=torch.float32, device=dev)
        K.gemm(q, k, S, M=1, N=n_keys, K=dk, lda=ldq, ldb=ldk, ldc=ldS, batch=Z, zdiv=H, a_s=(ldq, dk), b_s=(kv_sb, dk),
               c_s=(H * ldS, ldS))
        P = torch.empty(Z, 1, ldS, dtype=dt, device=dev)
        K.attn_softmax_fwd(S, ldS, None, 0, P, ldS, Z, H, 1, n_keys, scale, key_lens, False, False, None, None)
        K.gemm(P, v, O, M=1, N=dk, K=n_keys, lda=ldS, ldb=ldk, ldc=d, b_kmajor=True, batch=Z, zdiv=H, a_s=(H * ldS, ldS),
               b_s=(kv_sb, dk), c_s=(d, dk))
    y = torch.empty(Bb, d, dtype=dt, device=dev)
    K.gemm(O, cw(prm["o_w"]), y, M=Bb, N=d, K=d, lda=d, ldb=d, ldc=d, bias=prm["o_b"].data, residual=residual, ldr=d)
    return y


def project_memory(mem, prm, rows):
    """k | v projections of the encoder memory for encoder-decoder attention, computed once per sentence batch:
    mem [rows, d] -> [rows, 2d]."""
    d = mem.shape[1]
    wkv = fused([prm["k_w"], prm["v_w"]], 2 * d, d)
    bkv = fused_master([prm["k_b"], prm["v_b"]], 2 * d)
    kv = torch.empty(rows, 2 * d, dtype=mem.dtype, device=mem.device)
    K.gemm(mem, wkv, kv, M=rows, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, bias=bkv)
    return kv


# ------------------------------------------------------------------------------------------------
# Conformer convolution module
# ------------------------------------------------------------------------------------------------
class ConvModuleFn(torch.autograd.Function):
    """out = residual + mask(pw2(act(BN(dwconv(GLU(pw1(x)))))))   (modules/convolution.py:76-120).
    ``x`` is the conv_norm output with padded frames already zeroed (LayerNormFn with lens)."""

    @staticmethod
    def forward(ctx, x, residual, prm, bn_buf, act, B, T, lens, training, momentum, train, drop_o, ln_g=None, ln_b=None):
        M, d = x.shape
        dt, dev = x.dtype, x.device
        Kw = prm["dw_w"].shape[-1]
        w1 = cw(prm["pw1_w"]).view(2 * d, d)
        z = torch.empty(M, 2 * d, dtype=dt, device=dev) if train else None
        g = torch.empty(M, d, dtype=dt, device=dev)
        ctx.ln = None
        if ln_g is not None:
            # x is the block input BEFORE conv_norm (conv_module() only routes the row-block case here): LayerNorm and the
            # padded-frame mask of the module input (convolution.py:86-88) ride in the projection kernel's prologue
            x_pre, residual = x, x
            x = torch.empty_like(x_pre) if train else x_pre
            ln_mean = torch.empty(M, dtype=torch.float32, device=dev) if train else None
            ln_rstd = torch.empty(M, dtype=torch.float32, device=dev) if train else None
            pre = getattr(x_pre, "_s2t_chain", None)
            if pre is not None and pre["key"] == (w1.data_ptr(), ln_g.data_ptr(), bool(train), id(lens), T):
                # produced by the launch that made x_pre (attention(): s2t_rowblock_chain)
                g, z, ln_mean, ln_rstd = pre["g"], pre["z"], pre["mean"], pre["rstd"]
                x = pre["x"] if train else x_pre
                x_pre._s2t_chain = None
            else:
                K.rowblock_gemm(x_pre, w1, g, N=2 * d, ldc=d, act="glu", preact=z, ldp=2 * d, ln=(ln_g.data, ln_b.data),
                                ln_lens=lens, ln_T=T, x_ln=x if train else None, ln_stats=(ln_mean, ln_rstd) if train else None)
            ctx.ln = (ln_g, ln_b, getattr(x_pre, "_s2t_drop_o", None))
            ctx.ln_saved = (x_pre, ln_mean, ln_rstd)
        else:
            K.gemm(x, w1, g, M=M, N=2 * d, K=d, lda=d, ldb=d, ldc=d, act="glu", preact=z, ldp=2 * d, rows=lens)
        scale = shift = None
        packed = K.rows_geom(lens) is not None  # (s2t_amd/rows.py: utterance b's frames and halo rows from cu[b])
        wd = prm["dw_w"].data.view(d, Kw)
        a = torch.empty(M, d, dtype=dt, device=dev)
        D = mean = rstd = None
        if training:
            scale = torch.empty(d, dtype=torch.float32, device=dev)
            shift = torch.empty(d, dtype=torch.float32, device=dev)
            D = torch.empty(M, d, dtype=dt, device=dev)
            stats = torch.empty(K.dwconv_stat_partials(B, T), 2, d, dtype=torch.float32, device=dev)
            # (packed batch: the statistics cover the utterances' frames and halo rows; the padded frames behind them are zero
            # in the reference and add nothing to either sum, the divisor stays B * T)
            K.dwconv_fwd(g, wd, D, B, T, d, Kw, stats=stats, lens=lens if packed else None)
            mean = torch.empty(d, dtype=torch.float32, device=dev)
            rstd = torch.empty(d, dtype=torch.float32, device=dev)
            K.bn_finalize(stats, M, prm["bn_w"].data, prm["bn_b"].data, bn_buf["running_mean"], bn_buf["running_var"],
                          momentum, 1e-5, True, scale, shift, mean, rstd, d)
        rb2 = _rb_ok(a, d) and (residual is None or (residual.stride(0) % 8 == 0 and residual.stride(1) == 1))
        # eval: conv + BatchNorm on the running statistics + activation + mask in pointwise conv 2's row-block prologue
        conv_fused = (not training and rb2 and _CONV_EVAL_FUSED and Kw == 15 and T >= 18 and wd.dtype == torch.float32
                      and wd.is_contiguous())
        if not training and not conv_fused:  # ... or in one launch of their own
            K.dwconv_bn_eval_fwd(g, wd, a, B, T, d, Kw, prm["bn_w"].data, prm["bn_b"].data, bn_buf["running_mean"],
                                 bn_buf["running_var"], 1e-5, act, lens)
        y = torch.empty(M, d, dtype=dt, device=dev)
        fused_bn = training and rb2 and _BN_IN_PW2
        if training and not fused_bn:
            K.bn_act_fwd(D, a, scale, shift, act, M, d, lens, T)
        if fused_bn:
            # the BatchNorm apply + activation + mask ride in pointwise conv 2's prologue, which also writes their result
            # (the operand of pw2's weight gradient): one launch and one pass over D fewer
            K.rowblock_gemm(D, cw(prm["pw2_w"]).view(d, d), y, N=d, ldc=d, residual=residual,
                            ldr=residual.stride(0) if residual is not None else 0, row_lens=lens, row_T=T, drop=drop_o,
                            pre=(scale, shift, act), ln_lens=lens, ln_T=T, x_ln=a)
        elif conv_fused:
            K.rowblock_gemm(g, cw(prm["pw2_w"]).view(d, d), y, N=d, ldc=d, residual=residual,
                            ldr=residual.stride(0) if residual is not None else 0, row_lens=lens, row_T=T,
                            pre=(prm["bn_w"].data, prm["bn_b"].data, act), ln_lens=lens, ln_T=T,
                            conv=(wd, T, bn_buf["running_mean"], bn_buf["running_var"], 1e-5))
        elif rb2:
            K.rowblock_gemm(a, cw(prm["pw2_w"]).view(d, d), y, N=d, ldc=d, residual=residual,
                            ldr=residual.stride(0) if residual is not None else 0, row_lens=lens, row_T=T, drop=drop_o)
        else:
            K.gemm(a, cw(prm["pw2_w"]).view(d, d), y, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, residual=residual, ldr=d,
                   row_lens=lens, row_T=T, drop=drop_o)
        ctx.drop_o = drop_o
        if train:
            assert training, "gradients through the convolution module need training-mode BatchNorm"
            ctx.save_for_backward(x, z, g, D, a, scale, shift, mean, rstd)
        ctx.prm, ctx.act, ctx.dims, ctx.lens, ctx.has_res = prm, act, (B, T, d, Kw), lens, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, g, D, a, scale, shift, mean, rstd = ctx.saved_tensors
        prm = ctx.prm
        B, T, d, Kw = ctx.dims
        M = B * T
        dt, dev = x.dtype, x.device
        dres = dy.contiguous()
        dy = _drop_rows(dres, ctx.drop_o)
        # pw2 (a's padded rows are zero, so the weight gradient needs no extra mask; dA's are zeroed in bn_act_bwd)
        rows = ctx.lens if K.rows_geom(ctx.lens) is not None else None
        dA = _dgrad_rowblock(dy, prm["pw2_w"], d, rows=rows) if d == 256 else None
        if dA is None:
            dA = torch.empty(M, d, dtype=dt, device=dev)
            K.gemm(dy, cw(prm["pw2_w"]).view(d, d), dA, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_kmajor=True, rows=rows)
        _wgrad(dy, a, prm["pw2_w"].grad.view(d, d), d, d, M, d, d, rows=rows)
        # here dy rows of padded frames must not reach pw2's weight gradient: a is zero there -> contributes nothing
        sums = torch.empty(2 * d, dtype=torch.float32, device=dev)
        wd = prm["dw_w"].data.view(d, Kw)
        dZ = torch.empty(M, 2 * d, dtype=dt, device=dev)
        if _CONV_BWD_FUSED and dt == torch.bfloat16 and Kw <= 31:
            # BatchNorm reduce + fold, then ONE launch for the apply pass, the depthwise dgrad, the GLU backward and the
            # depthwise weight-gradient partials (dD and dG never reach HBM)
            K.bn_act_bwd(D, dA, None, scale, shift, mean, rstd, sums, M, ctx.act, M, d, ctx.lens, T,
                         dgamma=prm["bn_w"].grad, dbeta=prm["bn_b"].grad)
            slot = len(_DWQ["entries"]) if (_FOLD_DEFER and len(_DWQ["entries"]) < _FOLD_CAP and _arm_backward_end()) else None
            dwg = prm["dw_w"].grad.view(d, Kw)
            ws_dw, rows_dw = K.conv_bwd_fused(D, dA, g, z, wd, scale, shift, mean, rstd, sums, M, ctx.act, ctx.lens, dZ, dwg,
                                              B, T, d, Kw, defer_slot=slot)
            if slot is not None:
                _DWQ["entries"].append((ws_dw, dwg, rows_dw, d * Kw))
        else:
            if rows is not None:
                raise NotImplementedError("packed rows: the convolution module's backward runs through s2t_conv_bwd_fused only")
            dD = torch.empty(M, d, dtype=dt, device=dev)
            K.bn_act_bwd(D, dA, dD, scale, shift, mean, rstd, sums, M, ctx.act, M, d, ctx.lens, T,
                         dgamma=prm["bn_w"].grad, dbeta=prm["bn_b"].grad)
            dG = torch.empty(M, d, dtype=dt, device=dev)
            K.dwconv_fwd(dD, wd, dG, B, T, d, Kw, flip=True)
            K.dwconv_bwd_weight(g, dD, prm["dw_w"].grad.view(d, Kw), B, T, d, Kw)
            K.glu_bwd(z, dG, dZ, M, d)
        _wgrad(dZ, x, prm["pw1_w"].grad.view(2 * d, d), 2 * d, d, M, 2 * d, d, rows=rows)
        _ready(prm["pw1_w"], prm["dw_w"], prm["bn_w"], prm["bn_b"], prm["pw2_w"])
        if ctx.ln is not None:
            ln_g, ln_b, up_drop = ctx.ln
            x_pre, ln_mean, ln_rstd = ctx.ln_saved
            dxp = _dgrad_ln_backward(dZ, prm["pw1_w"], 2 * d, x_pre, ln_g, ln_b, ln_mean, ln_rstd, ctx.lens, T, dres, up_drop)
            if dxp is not None:
                return (dxp, None) + (None,) * 12
        dx = torch.empty(M, d, dtype=dt, device=dev)
        K.gemm(dZ, cw(prm["pw1_w"]).view(2 * d, d), dx, M=M, N=d, K=2 * d, lda=2 * d, ldb=d, ldc=d, b_kmajor=True, rows=rows)
        if ctx.ln is not None:
            ln_g, ln_b, up_drop = ctx.ln
            x_pre, ln_mean, ln_rstd = ctx.ln_saved
            dxp = _ln_backward(x_pre, ln_g, ln_b, dx, ln_mean, ln_rstd, ctx.lens, T, dres, up_drop)
            return (dxp, None) + (None,) * 12
        return (dx, (dres if ctx.has_res else None)) + (None,) * 12


_CONV_EVAL_FUSED = os.environ.get("S2T_CONV_EVAL_FUSED", "1") != "0"  # eval: depthwise conv + BatchNorm + activation in pw2's prologue
_CONV_BWD_FUSED = os.environ.get("S2T_CONV_BWD_FUSED", "1") != "0"  # s2t_conv_bwd_fused in ConvModuleFn.backward (bf16)


def conv_module(x, residual, prm, bn_buf, act, B, T, lens, training, momentum=0.1, p_out=0.0, ln=None):
    """``ln`` = (gamma, beta) of conv_norm: ``x`` is then the block input before that LayerNorm and doubles as the residual
    (``residual`` must be None); the LayerNorm and the padded-frame mask ride in the row-block kernel where it applies."""
    if ln is not None:
        assert residual is None
        if not _rb_ok(x, 2 * x.shape[1], "glu"):
            y, xr = layer_norm(x, ln[0], ln[1], lens, T, fork=True)
            return conv_module(y, xr, prm, bn_buf, act, B, T, lens, training, momentum, p_out)
    drop_o = DROPOUT.next(p_out if training else 0.0, x.device)
    lg, lb = ln if ln is not None else (None, None)
    return _tag_drop(ConvModuleFn.apply(x, residual, prm, bn_buf, act, B, T, lens, training, momentum,
                                        torch.is_grad_enabled(), drop_o, lg, lb), drop_o)


# ------------------------------------------------------------------------------------------------
# PDS multi-scale fusion: depthwise (kernel = stride = r) conv + BatchNorm + activation
# ------------------------------------------------------------------------------------------------
class PoolBnActFn(torch.autograd.Function):
    """a = act(BN(bias + sum_k x[t*r + k] * w[k])) on channels-last rows: the middle of DownSampleConvolutionModule
    (fairseq/modules/downsample_convolution.py:97-106).  BatchNorm statistics over all B*T' rows, padded ones included,
    like nn.BatchNorm1d on the (B, C, T') tensor."""

    @staticmethod
    def forward(ctx, x, prm, bn_buf, act, B, T, r, training, momentum, train):
        M, d = x.shape
        dt, dev = x.dtype, x.device
        To = T // r
        Mo = B * To
        wd = prm["dw_w"].data.view(d, r)
        D = torch.empty(Mo, d, dtype=dt, device=dev)
        scale = torch.empty(d, dtype=torch.float32, device=dev)
        shift = torch.empty(d, dtype=torch.float32, device=dev)
        mean = rstd = None
        if training:
            stats = torch.empty(K.dwpool_stat_partials(B, To), 2, d, dtype=torch.float32, device=dev)
            K.dwpool_fwd(x, wd, prm["dw_b"].data, D, B, T, d, r, stats=stats)
            mean = torch.empty(d, dtype=torch.float32, device=dev)
            rstd = torch.empty(d, dtype=torch.float32, device=dev)
            K.bn_finalize(stats, Mo, prm["bn_w"].data, prm["bn_b"].data, bn_buf["running_mean"], bn_buf["running_var"],
                          momentum, 1e-5, True, scale, shift, mean, rstd, d)
        else:
            K.dwpool_fwd(x, wd, prm["dw_b"].data, D, B, T, d, r)
            K.bn_finalize(None, 0, prm["bn_w"].data, prm["bn_b"].data, bn_buf["running_mean"], bn_buf["running_var"],
                          momentum, 1e-5, False, scale, shift, None, None, d)
        a = torch.empty(Mo, d, dtype=dt, device=dev)
        K.bn_act_fwd(D, a, scale, shift, act, Mo, d)
        if train:
            assert training, "gradients through the fusion convolution need training-mode BatchNorm"
            ctx.save_for_backward(x, D, scale, shift, mean, rstd)
        ctx.prm, ctx.act, ctx.dims = prm, act, (B, T, To, d, r)
        return a

    @staticmethod
    def backward(ctx, dA):
        x, D, scale, shift, mean, rstd = ctx.saved_tensors
        prm = ctx.prm
        B, T, To, d, r = ctx.dims
        Mo = B * To
        dt, dev = x.dtype, x.device
        dD = torch.empty(Mo, d, dtype=dt, device=dev)
        sums = torch.empty(2 * d, dtype=torch.float32, device=dev)
        K.bn_act_bwd(D, dA.contiguous(), dD, scale, shift, mean, rstd, sums, Mo, ctx.act, Mo, d,
                     dgamma=prm["bn_w"].grad, dbeta=prm["bn_b"].grad)
        dx = (torch.zeros if To * r != T else torch.empty)(B * T, d, dtype=dt, device=dev)
        K.dwpool_bwd(x, prm["dw_w"].data.view(d, r), dD, dx, prm["dw_w"].grad.view(d, r), prm["dw_b"].grad, B, T, d, r)
        _ready(prm["dw_w"], prm["dw_b"], prm["bn_w"], prm["bn_b"])
        return dx, None, None, None, None, None, None, None, None, None


def pool_bn_act(x, prm, bn_buf, act, B, T, r, training, momentum=0.1):
    return PoolBnActFn.apply(x, prm, bn_buf, act, B, T, r, training, momentum, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# Conv1d subsampler
# ------------------------------------------------------------------------------------------------
def _conv_out_len(T):
    return (T - 1) // 2 + 1


_SUB_TAPS = os.environ.get("S2T_SUB_TAPS", "0") == "1"  # subsampler input gradient tap by tap (five accumulating products)


class SubsampleFn(torch.autograd.Function):
    """2 x [Conv1d(k=5, stride 2, pad 2) -> GLU]  (modules/speech_to_text/subsampling.py:106-159) as overlapping-row
    GEMMs over zero-padded (B, T+pad, C) buffers; the padded-frame mask of the encoder (s2t_transformer.py:1765)
    is fused into the second GEMM's epilogue.  Weights are stored [Cout][k][Cin] (see Conv1dSubsampling)."""

    @staticmethod
    def forward(ctx, src, w0, b0, w1, b1, out_lens, dt, train):
        B, T, Cin = src.shape
        dev = src.device
        kk = 5
        C0, C1 = w0.shape[0], w1.shape[0]  # conv output channels (before GLU)
        T1, = (_conv_out_len(T),)
        T2 = _conv_out_len(T1)
        Cin_p = _pad8(Cin)
        assert Cin_p == Cin, "input feature dim must be a multiple of 8"
        # (a tail of k rows behind the last utterance: the weight-gradient product reads every buffer as overlapping rows
        # of k * C elements at a stride of 2 * C, junk rows of the last utterance included, see backward)
        Tp0 = 2 * T1 + 4
        xp = torch.zeros(B * Tp0 + kk, Cin, dtype=dt, device=dev)[:B * Tp0].view(B, Tp0, Cin)
        xp[:, 2:2 + T].copy_(src)
        H0 = C0 // 2
        Tp1 = 2 * T2 + 4
        y1p = torch.zeros(B * Tp1 + kk, H0, dtype=dt, device=dev)[:B * Tp1].view(B, Tp1, H0)
        z1 = torch.empty(B, T1, C0, dtype=dt, device=dev) if train else None
        K.gemm(xp, cw(w0).view(C0, kk * Cin), y1p[:, 2:], M=T1, N=C0, K=kk * Cin, lda=2 * Cin, ldb=kk * Cin, ldc=H0,
               batch=B, a_s=(Tp0 * Cin, 0), c_s=(Tp1 * H0, 0), bias=b0.data, act="glu", preact=z1, ldp=C0,
               p_s=(T1 * C0, 0))
        H1 = C1 // 2
        y2 = torch.empty(B, T2, H1, dtype=dt, device=dev)
        z2 = torch.empty(B, T2, C1, dtype=dt, device=dev) if train else None
        K.gemm(y1p, cw(w1).view(C1, kk * H0), y2, M=T2, N=C1, K=kk * H0, lda=2 * H0, ldb=kk * H0, ldc=H1, batch=B,
               a_s=(Tp1 * H0, 0), c_s=(T2 * H1, 0), bias=b1.data, act="glu", preact=z2, ldp=C1, p_s=(T2 * C1, 0),
               row_lens=out_lens, row_T=T2)
        if train:
            ctx.save_for_backward(xp, y1p, z1, z2)
        ctx.p = (w0, b0, w1, b1)
        ctx.dims = (B, T, Cin, C0, C1, T1, T2, Tp0, Tp1)
        ctx.out_lens = out_lens
        return y2.view(B * T2, H1)

    @staticmethod
    def backward(ctx, dy):
        xp, y1p, z1, z2 = ctx.saved_tensors
        w0, b0, w1, b1 = ctx.p
        B, T, Cin, C0, C1, T1, T2, Tp0, Tp1 = ctx.dims
        kk = 5
        H0, H1 = C0 // 2, C1 // 2
        dt, dev = xp.dtype, xp.device
        dy = dy.contiguous()
        # Weight gradients as ONE product over all utterances: the padded input buffers hold Tp = 2 (T' + 2) rows per
        # utterance, so with TWO extra all-zero gradient rows behind every utterance's T' the overlapping im2col rows
        # (row r of the product = buffer offset r * 2 * C, length k * C) of all utterances form a single matrix with a
        # uniform row stride — K = B (T' + 2) instead of B separate K = T' products accumulated with atomics; the two junk
        # rows per utterance multiply zeros.  The products then join the grouped weight-gradient launch (bias included).
        assert Tp1 == 2 * (T2 + 2) and Tp0 == 2 * (T1 + 2)
        # layer 2: GLU backward (padded frames carry no gradient)
        # (two all-zero rows in FRONT of the first utterance as well: the input-gradient products below read every utterance's rows
        # from two rows before its first)
        dz2_buf = torch.empty(2 + B * (T2 + 2), C1, dtype=dt, device=dev)
        dz2_buf[:2].zero_()
        dz2 = dz2_buf[2:].view(B, T2 + 2, C1)
        dz2[:, T2:].zero_()
        K.glu_bwd(z2.view(B * T2, C1), dy, dz2, B * T2, H1, ctx.out_lens, T2, out_pad=2)
        _wgrad(dz2.view(B * (T2 + 2), C1), y1p, w1.grad.view(C1, kk * H0), C1, kk * H0, B * (T2 + 2), C1, 2 * H0, 1.0,
               b1.grad)
        # d y1 (the layer-1 output frames; frame f is row f + 2 of the padded buffer the forward convolved): frame f meets
        # (t, tap) with 2 t + tap = f + 2 — the EVEN frames f = 2 u the taps 4, 2, 0 of output frames u-1, u, u+1, the ODD frames
        # f = 2 u + 1 the taps 3, 1 of u, u+1.  The output frames of a product row are consecutive rows of dz2 (the zero rows
        # between the utterances serve as the frames before 0 and beyond T'), so each parity is ONE product over overlapping
        # operand rows (stride one row) against its taps' weights stacked along K, written straight into the contiguous
        # gradient: two launches that write every row once, where five tap-by-tap products accumulated into a padded buffer
        # (5 x read-modify-write of 33 MB at 64 x 1000 frames, a zero fill and a copy of the frames out of it): -0.07 ms per step
        w1c = cw(w1).view(C1, kk, H0)
        if _SUB_TAPS:  # (A/B switch: the tap-by-tap form)
            dy1p = torch.zeros(B, Tp1, H0, dtype=dt, device=dev)
            for tap in range(kk):
                out = dy1p.view(-1)[tap * H0:]
                K.gemm(dz2, w1c[:, tap], out, M=T2, N=H0, K=C1, lda=C1, ldb=kk * H0, ldc=2 * H0, b_kmajor=True, batch=B,
                       a_s=((T2 + 2) * C1, 0), c_s=(Tp1 * H0, 0), residual=out, ldr=2 * H0)
            dy1 = dy1p[:, 2:2 + T1].contiguous().view(B * T1, H0)
        else:
            dy1 = torch.empty(B * T1, H0, dtype=dt, device=dev)
            for parity, taps in ((0, (4, 2, 0)), (1, (3, 1))):
                nt = len(taps)
                rows_p = (T1 + 1 - parity) // 2
                if rows_p == 0:
                    continue
                wcat = torch.cat([w1c[:, tap] for tap in taps], 0)  # [nt * C1, H0], k-major
                K.gemm(dz2_buf[4 - nt:], wcat, dy1.view(-1)[parity * H0:], M=rows_p, N=H0, K=nt * C1, lda=C1, ldb=H0, ldc=2 * H0,
                       b_kmajor=True, batch=B, a_s=((T2 + 2) * C1, 0), c_s=(T1 * H0, 0))
        # layer 1
        dz1 = torch.empty(B, T1 + 2, C0, dtype=dt, device=dev)
        dz1[:, T1:].zero_()
        K.glu_bwd(z1.view(B * T1, C0), dy1, dz1, B * T1, H0, None, T1, out_pad=2)
        _wgrad(dz1.view(B * (T1 + 2), C0), xp, w0.grad.view(C0, kk * Cin), C0, kk * Cin, B * (T1 + 2), C0, 2 * Cin, 1.0,
               b0.grad)
        _ready(w0, b0, w1, b1)
        return None, None, None, None, None, None, None, None


def subsample(src, w0, b0, w1, b1, out_lens, dt):
    return SubsampleFn.apply(src, w0, b0, w1, b1, out_lens, dt, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# Embedding
# ------------------------------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    """x = scale * E[tokens] + sinusoid[pos]  (models/transformer.py:1304-1323)."""

    @staticmethod
    def forward(ctx, tokens, pos, E, tab, scale, pad_idx):
        n = tokens.numel()
        d = E.shape[1]
        out = torch.empty(n, d, dtype=cw(E).dtype, device=tokens.device)
        K.embedding_fwd(tokens, pos, cw(E), tab, out, n, d, scale)
        ctx.save_for_backward(tokens)
        ctx.E, ctx.scale, ctx.pad_idx = E, scale, pad_idx
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        E = ctx.E
        dout = dout.contiguous()
        with _on_wgrad_stream(tokens, dout):  # the table may be tied to a projection whose wgrad runs on that stream
            K.embedding_bwd(tokens, dout, E.grad, tokens.numel(), E.shape[1], ctx.scale, ctx.pad_idx)
        _ready(E)
        return None, None, None, None, None, None


def embedding(tokens, pos, E, tab, scale, pad_idx):
    return EmbeddingFn.apply(tokens, pos, E, tab, scale, pad_idx)


# ------------------------------------------------------------------------------------------------
# Losses
# ------------------------------------------------------------------------------------------------
class LabelSmoothedCEFn(torch.autograd.Function):
    """criterions/label_smoothed_cross_entropy.py:42-60 — returns (loss, nll, n_correct, total) as a 4-vector;
    the gradient w.r.t. the logits is produced in the same pass."""

    @staticmethod
    def forward(ctx, logits, target, eps, pad_idx, train, bound=None):
        rows, V = logits.shape
        assert logits.stride(1) == 1
        sums = torch.zeros(4, dtype=torch.float32, device=logits.device)
        dl = None
        if train:
            dl = torch.empty(rows, _pad8(V), dtype=logits.dtype, device=logits.device)[:, :V]
        K.ls_cross_entropy(logits, logits.stride(0), rows, V, target, pad_idx, eps, dl, dl.stride(0) if train else 0, sums,
                           bound=bound)
        if train:
            ctx.save_for_backward(dl)
        return sums

    @staticmethod
    def backward(ctx, dsums):
        (dl,) = ctx.saved_tensors
        # only sums[0] (the loss) is differentiable
        return dl * dsums[0].to(dl.dtype), None, None, None, None, None


def label_smoothed_ce(logits, target, eps, pad_idx, rows=None):
    """``rows``: the lengths tensor of a packed batch whose rows ``logits`` / ``target`` hold (only the live rows are read)."""
    return LabelSmoothedCEFn.apply(logits, target, eps, pad_idx, torch.is_grad_enabled() and logits.requires_grad,
                                   rows if K.rows_geom(rows) is not None else None)


_SIDE = {"streams": {}, "pending": []}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE["streams"]:
        _SIDE["streams"][key] = torch.cuda.Stream(device=device)
    return _SIDE["streams"][key]


def join_side_streams():
    """Make the current stream wait for every side-stream branch opened since the last join (``ctc_loss(side=True)``)."""
    cur = torch.cuda.current_stream()
    for st, _keep in _SIDE["pending"]:
        cur.wait_stream(st)
    _SIDE["pending"] = []  # the buffers kept alive for the side branches may now return to the allocator


class CTCLossFn(torch.autograd.Function):
    """sum_b CTC nll_b with zero_infinity (criterions/ctc.py:243-245,435-474); logits are batch-major [B*T, V].

    ``side=True``: the forward kernels (row logsumexp, alpha/beta: one or two workgroups per utterance walking T' dependent
    steps, i.e. a long thin launch that leaves most CUs idle) are issued on a side stream so that they run beside whatever
    the caller launches next (the decoder); every buffer is allocated on the calling stream first, so allocator lifetimes
    stay ordered on it, and the caller must ``join_side_streams()`` before it consumes the returned loss."""

    @staticmethod
    def forward(ctx, logits, B, T, targets, tgt_lens, in_lens, blank, side, rows=None):
        V = logits.shape[1]
        dev = logits.device
        S = targets.shape[1]
        Lmax = 2 * S + 1
        lse = torch.empty(B * T, dtype=torch.float32, device=dev)
        assert logits.stride(1) == 1
        ld = logits.stride(0)
        alpha = torch.empty(B, T, Lmax, dtype=torch.float32, device=dev)
        beta = torch.empty(B, T, Lmax, dtype=torch.float32, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        clean = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)

        def run():
            K.argmax_lse(logits, ld, B * T, V, None, None, lse, bound=rows)
            K.ctc_loss_fwd(logits, ld, B, T, V, lse, targets, S, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, rows=rows)
            torch.nan_to_num(nll, nan=float("nan"), posinf=0.0, out=clean)  # zero_infinity
            torch.sum(clean, dim=0, out=out)

        if side:
            st = _side_stream(dev)
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                run()
            # every buffer the branch touches was allocated on the calling stream: it must not go back to that stream's
            # allocator (e.g. under no_grad, where nothing is saved for backward) before the join
            _SIDE["pending"].append((st, (logits, lse, alpha, beta, nll, clean, out, targets, tgt_lens, in_lens)))
        else:
            run()
        ctx.save_for_backward(logits, lse, alpha, beta, nll, targets, tgt_lens, in_lens)
        ctx.dims = (B, T, V, S, Lmax, blank)
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, g):
        logits, lse, alpha, beta, nll, targets, tgt_lens, in_lens = ctx.saved_tensors
        B, T, V, S, Lmax, blank = ctx.dims
        grad = torch.empty(B * T, _pad8(V), dtype=logits.dtype, device=logits.device)[:, :V]
        gs = g.detach().reshape(1).float().contiguous()  # upstream gradient of the summed loss, stays on the device
        K.ctc_loss_bwd(logits, logits.stride(0), B, T, V, lse, targets, S, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll,
                       1.0, grad, grad.stride(0), gscale_dev=gs, rows=ctx.rows)
        return grad, None, None, None, None, None, None, None, None


def ctc_loss(logits, B, T, targets, tgt_lens, in_lens, blank=0, side=False, rows=None):
    """``rows``: the lengths tensor of a packed batch (s2t_amd/rows.py) whose rows ``logits`` holds."""
    return CTCLossFn.apply(logits, B, T, targets, tgt_lens, in_lens, blank, bool(side),
                           rows if K.rows_geom(rows) is not None else None)


# ------------------------------------------------------------------------------------------------
# Strided Conv1d (k odd, pad (k-1)/2) over time as an overlapping-row GEMM — PDS down-sampling
# ------------------------------------------------------------------------------------------------
class Conv1dFn(torch.autograd.Function):
    """y[b,t,:] = bias + sum_tap W[:, tap, :] x[b, stride*t + tap - pad, :]  (nn.Conv1d on (B,C,T), channels-last here).
    Reference: Downsampling.conv in models/speech_to_text/pdss2t_transformer.py:72-90,127-129.
    ``x`` is [B*T, Cin] with padded frames already zeroed; weight is stored [Cout][k][Cin]."""

    @staticmethod
    def forward(ctx, x, w, b, B, T, stride, train):
        Cout, kk, Cin = w.shape
        pad = (kk - 1) // 2
        dt, dev = x.dtype, x.device
        Tout = (T + 2 * pad - kk) // stride + 1
        Tp = stride * (Tout - 1) + kk  # rows touched
        Tp = max(Tp, T + pad) + 1
        xp = torch.zeros(B, Tp, Cin, dtype=dt, device=dev)
        xp[:, pad:pad + T].copy_(x.view(B, T, Cin))
        y = torch.empty(B * Tout, Cout, dtype=dt, device=dev)
        K.gemm(xp, cw(w).view(Cout, kk * Cin), y, M=Tout, N=Cout, K=kk * Cin, lda=stride * Cin, ldb=kk * Cin, ldc=Cout,
               batch=B, a_s=(Tp * Cin, 0), c_s=(Tout * Cout, 0), bias=b.data)
        if train:
            ctx.save_for_backward(xp)
        ctx.p, ctx.dims = (w, b), (B, T, Tout, Tp, Cin, Cout, kk, pad, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xp,) = ctx.saved_tensors
        w, b = ctx.p
        B, T, Tout, Tp, Cin, Cout, kk, pad, stride = ctx.dims
        dt, dev = xp.dtype, xp.device
        dy = dy.contiguous()
        K.gemm(dy, xp, w.grad.view(Cout, kk * Cin), M=Cout, N=kk * Cin, K=Tout, lda=Cout, ldb=stride * Cin, ldc=kk * Cin,
               a_kmajor=True, b_kmajor=True, batch=B, a_s=(Tout * Cout, 0), b_s=(Tp * Cin, 0), c_s=(0, 0), c_atomic=True)
        K.colsum_accum(dy, Cout, b.grad, B * Tout, Cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dxp = torch.zeros(B, Tp, Cin, dtype=dt, device=dev)
            wc = cw(w).view(Cout, kk * Cin)
            for tap in range(kk):
                out = dxp.view(-1)[tap * Cin:]
                K.gemm(dy, wc[:, tap * Cin:], out, M=Tout, N=Cin, K=Cout, lda=Cout, ldb=kk * Cin, ldc=stride * Cin,
                       b_kmajor=True, batch=B, a_s=(Tout * Cout, 0), c_s=(Tp * Cin, 0), residual=out, ldr=stride * Cin)
            dx = dxp[:, pad:pad + T].contiguous().view(B * T, Cin)
        _ready(w, b)
        return dx, None, None, None, None, None, None


def conv1d(x, w, b, B, T, stride):
    return Conv1dFn.apply(x, w, b, B, T, stride, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# CTC-guided compression of the frame axis (s2t_transformer.py:1948-1986)
# ------------------------------------------------------------------------------------------------
class CompressRowsFn(torch.autograd.Function):
    """y[b][j] = x[b][src[b][j]] for j < new_lens[b], zero rows beyond; backward scatters the rows back."""

    @staticmethod
    def forward(ctx, x, src, new_lens, B, T, Tn):
        d = x.shape[1]
        y = torch.empty(B * Tn, d, dtype=x.dtype, device=x.device)
        K.compress_rows(x.contiguous(), y, src, new_lens, B, T, Tn, d)
        ctx.src, ctx.new_lens, ctx.dims = src, new_lens, (B, T, Tn, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, Tn, d = ctx.dims
        dx = torch.zeros(B * T, d, dtype=dy.dtype, device=dy.device)
        K.compress_rows(dy.contiguous(), dx, ctx.src, ctx.new_lens, B, T, Tn, d, scatter=True)
        return dx, None, None, None, None, None


def ctc_compress_plan(logit2d, lens32, B, T, blank, threshold):
    """-> (src [B, T] int32, new_lens [B] int32): frames kept = blank posterior < threshold (fp32 softmax)."""
    V = logit2d.shape[1]
    dev = logit2d.device
    lse = torch.empty(B * T, dtype=torch.float32, device=dev)
    K.argmax_lse(logit2d, logit2d.stride(0), B * T, V, None, None, lse)
    src = torch.zeros(B, T, dtype=torch.int32, device=dev)
    new_lens = torch.empty(B, dtype=torch.int32, device=dev)
    K.ctc_compress_plan(logit2d, lse, lens32, B, T, blank, threshold, src, new_lens)
    return src, new_lens


# ------------------------------------------------------------------------------------------------
# SATE adapter (inter_league)
# ------------------------------------------------------------------------------------------------
_POS32 = {}
# s2t_relpos_glue walks the position rows in chunks of 512 (round 5): any length; S2T_GLUE_MAX_T=256 restores the round-4 routing
# (longer sequences through s2t_relpos_dqv + the split-K position-table GEMM, padded rows only) for A/B measurements
_GLUE_MAX_T = int(os.environ.get("S2T_GLUE_MAX_T", "32768"))
_ATTN_ONE_PASS = os.environ.get("S2T_ATTN_ONE_PASS", "1") != "0"  # s2t_attn_bwd_one_pass: plain attention, dq / dk / dv in one launch
_ATTN_ONE_PASS_MIN_TK = int(os.environ.get("S2T_ATTN_ONE_PASS_MIN_TK", "160"))  # (keys per utterance from which it beats the two kernels)
_RELPOS_ONE_PASS = os.environ.get("S2T_RELPOS_ONE_PASS", "1") != "0"  # s2t_relpos_attn_bwd: T' <= 256, the whole rel-pos backward in one launch
_RELPOS_GLUE = os.environ.get("S2T_RELPOS_GLUE", "1") != "0"  # s2t_relpos_glue: (Q+v) branch + bias sums + position-table gradient in one pass over dbd
_DP_SPLIT = int(os.environ.get("S2T_DP_SPLIT", "16"))  # K split of the position-table gradient GEMM (M = 2T-1, N = 64 per head, K = B*T)
_POSW_SPLIT = int(os.environ.get("S2T_POSW_SPLIT", "0"))  # experiment: K split of the small fp32 linear_pos weight-gradient GEMM


def _pos_table_f32(pos_tab):
    """fp32 copy of a (cached, constant) relative-position table: made once, not once per layer and step."""
    if pos_tab.dtype == torch.float32:
        return pos_tab
    key = (pos_tab.data_ptr(), tuple(pos_tab.shape), str(pos_tab.device))
    hit = _POS32.get(key)
    if hit is None or hit[0] is not pos_tab:
        if len(_POS32) > 64:
            _POS32.clear()
        hit = _POS32[key] = (pos_tab, pos_tab.float())
    return hit[1]


class AdapterFn(torch.autograd.Function):
    """out = x + dist @ W_embed, dist = softmax(ctc_logit / tau)   (modules/speech_to_text/adapter.py:214-217,264-266,
    296-297); rows flagged by ``rows`` take the (optionally smoothed) one-hot distribution of ``oracle`` instead
    (adapter.py:245-262, the PAE ground-truth curriculum) and pass no gradient to the logits.
    ``packed``: the lengths tensor of a packed batch (s2t_amd/rows.py) whose rows x / logit hold: only the live rows are
    computed and enter the weight gradient."""

    @staticmethod
    def forward(ctx, x, logit, w, tau, train, oracle, rows, smooth, packed=None):
        M, d = x.shape
        V = w.shape[0]
        assert logit.stride(1) == 1
        dt, dev = x.dtype, x.device
        lg = logit if logit.dtype == dt else logit.to(dt)
        ldp = _pad8(V)
        P = torch.empty(M, ldp, dtype=dt, device=dev)
        K.row_softmax_fwd(lg, lg.stride(0), P, ldp, M, V, 1.0 / tau, bound=packed)
        if oracle is not None:
            assert packed is None, "the ground-truth curriculum of PAE runs on padded rows"
            on, off = (0.9 + 0.1 / V, 0.1 / V) if smooth else (1.0, 0.0)
            Pv = P[:, :V]
            Pv.masked_fill_(rows[:, None], off)
            idx = oracle.view(-1, 1)
            Pv.scatter_(1, idx, torch.where(rows[:, None], torch.full_like(Pv[:, :1], on), Pv.gather(1, idx)))
        y = torch.empty(M, d, dtype=dt, device=dev)
        K.gemm(P, cw(w), y, M=M, N=d, K=V, lda=ldp, ldb=d, ldc=d, b_kmajor=True, residual=x, ldr=d, rows=packed)
        if train:
            ctx.save_for_backward(P)
        ctx.w, ctx.tau, ctx.V, ctx.ldt = w, tau, V, logit.dtype
        ctx.rows = rows if oracle is not None else None
        ctx.packed = packed
        return y

    @staticmethod
    def backward(ctx, dy):
        (P,) = ctx.saved_tensors
        w, V = ctx.w, ctx.V
        M, ldp = P.shape
        d = w.shape[1]
        dt, dev = P.dtype, P.device
        dy = dy.contiguous()
        dP = torch.empty(M, ldp, dtype=dt, device=dev)
        K.gemm(dy, cw(w), dP, M=M, N=V, K=d, lda=d, ldb=d, ldc=ldp, rows=ctx.packed)
        dlogit = torch.empty(M, ldp, dtype=dt, device=dev)
        K.row_softmax_bwd(P, ldp, dP, ldp, dlogit, ldp, M, V, 1.0 / ctx.tau, bound=ctx.packed)
        if ctx.rows is not None:
            dlogit.masked_fill_(ctx.rows[:, None], 0.0)
        _wgrad(P, dy, w.grad, V, d, M, ldp, d, rows=ctx.packed)
        _ready(w)
        dl = dlogit[:, :V]
        return dy, (dl if ctx.ldt == dt else dl.to(ctx.ldt)), None, None, None, None, None, None, None


def adapter_inter_league(x, logit, w, tau=1.0, oracle=None, oracle_rows=None, oracle_smooth=False, rows=None):
    """x [M, d], logit [M, V]; oracle int64 [M] labels, oracle_rows bool [M] (both or neither); ``rows``: packed geometry."""
    return AdapterFn.apply(x, logit, w, tau, torch.is_grad_enabled(), oracle, oracle_rows, bool(oracle_smooth),
                           rows if K.rows_geom(rows) is not None else None)
