"""Thin, allocation-explicit Python wrappers over the C-ABI kernels (no autograd here)."""
import ctypes as C
from typing import Optional

import os

import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---- packed rows (include/s2t_hip.h, "Packed rows"; geometry object: s2t_amd/rows.py) ----------------------------------
# A lengths tensor of a packed batch carries its geometry as ``lens._pk`` (row map, cu, ...).  Wherever a wrapper takes a
# (lens, T) pair it hands the kernels the row map instead; ``rows`` (a lengths tensor as well) names the batch for launches
# that have no mask of their own and only need the live-row bound.
ROWS_PACKED, ROWS_BOUND = -1, -2


def rows_geom(lens):
    return getattr(lens, "_pk", None) if lens is not None else None


def _mask(lens, T, rows=None):
    """(pointer, T) of a (lens, T) argument pair: the uniform layout, the row map of a packed batch, or its bound only."""
    g = rows_geom(lens)
    if g is not None:
        return g.map_ptr, ROWS_PACKED
    if lens is not None:
        assert lens.dtype == torch.int32
        return lens.data_ptr(), int(T)
    g = rows_geom(rows)
    if g is not None:
        return g.map_ptr, ROWS_BOUND
    return None, 0


def _cu(lens):
    g = rows_geom(lens)
    return g.cu.data_ptr() if g is not None else None


def _live(rows):
    g = rows_geom(rows)
    return g.map_ptr - 4 if g is not None else None


def _prof_rows(M, *cands):
    """Rows a profiled launch really works on (bench.py's roofline leg only: a device-to-host copy of a packed batch's live
    row count)."""
    for c in cands:
        g = rows_geom(c)
        if g is not None:
            return min(M, g.live_rows())
    return M


def subsampled_lengths(src_lengths, Tp, len64, len32, mask, n_layers=2):
    """s2t_subsampled_lengths: the subsampler's output lengths (int64 and int32) and the padding mask [B, Tp] (torch.bool), in place."""
    L.require_cuda(src_lengths, len64, len32, mask)
    assert src_lengths.dtype == torch.int64 and src_lengths.is_contiguous() and mask.dtype == torch.bool and mask.is_contiguous()
    assert len64.dtype == torch.int64 and len32.dtype == torch.int32 and tuple(mask.shape) == (src_lengths.numel(), Tp)
    _call("s2t_subsampled_lengths", src_lengths.data_ptr(), src_lengths.numel(), Tp, n_layers, len64.data_ptr(), len32.data_ptr(),
          mask.data_ptr())


def token_positions(tokens, pad_idx, positions, counts):
    """s2t_token_positions: fairseq's make_positions (int32) and the non-pad count per row, in place."""
    L.require_cuda(tokens, positions, counts)
    B, U = tokens.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and positions.dtype == torch.int32 and positions.is_contiguous()
    assert counts.dtype == torch.int32 and tuple(positions.shape) == (B, U) and counts.numel() == B
    _call("s2t_token_positions", tokens.data_ptr(), B, U, int(pad_idx), positions.data_ptr(), counts.data_ptr())


def ctc_targets(target, pad_idx, eos_idx, tmat, counts):
    """s2t_ctc_targets: the left-packed label matrix and the label counts of criterions/ctc.py:516-540, in place."""
    L.require_cuda(target, tmat, counts)
    B, U = target.shape
    assert target.dtype == torch.int64 and target.is_contiguous() and tmat.dtype == torch.int64 and tmat.is_contiguous()
    assert counts.dtype == torch.int32 and tuple(tmat.shape) == (B, U) and counts.numel() == B and tmat.data_ptr() != target.data_ptr()
    _call("s2t_ctc_targets", target.data_ptr(), B, U, int(pad_idx), int(eos_idx), tmat.data_ptr(), counts.data_ptr())


def gather_rows_i64(src, row_map, U, fill, out):
    """s2t_gather_rows_i64: ``src`` [B * U] int64 through a packed batch's row map (``fill`` on rows that hold no token), in place."""
    L.require_cuda(src, row_map, out)
    assert src.dtype == torch.int64 and src.is_contiguous() and out.dtype == torch.int64 and out.is_contiguous()
    assert row_map.dtype == torch.int32 and row_map.is_contiguous() and out.numel() == row_map.numel()
    _call("s2t_gather_rows_i64", src.data_ptr(), row_map.data_ptr(), row_map.numel(), U, int(fill), out.data_ptr())


def rows_geometry(lens, B, T, halo, cu, buf):
    _call("s2t_rows_geometry", lens.data_ptr(), B, T, halo, cu.data_ptr(), buf.data_ptr())


def pack_rows(src, out, lens, to_packed=True):
    """s2t_pack_rows: padded [B*T, C] -> packed rows (or back: ``out`` zero-filled by the caller) of the batch ``lens`` names."""
    g = rows_geom(lens)
    L.require_cuda(src, out)
    assert src.dtype == out.dtype and src.is_contiguous() and out.is_contiguous() and src.shape == out.shape == (g.M, src.shape[1])
    _call("s2t_pack_rows", L.dtype_id(src.dtype), src.data_ptr(), out.data_ptr(), g.map_ptr, g.M, g.T, src.shape[1],
          int(bool(to_packed)))


# Optional per-launch timing of the GEMM symbols (bench.py's roofline leg): when a list is installed here every
# s2t_gemm launch is bracketed by HIP events on the launch stream and (symbol, flops, ev0, ev1) is appended.
GEMM_PROFILE = None


def gemm_symbol(args):
    """Name of the kernel s2t_gemm launches for ``args`` exactly as rocprofv3 prints it (s2t_gemm_describe)."""
    buf = C.create_string_buffer(160)
    L.check(L.lib().s2t_gemm_describe(C.byref(args), buf, 160), "s2t_gemm_describe")
    return buf.value.decode()


def gemm_configure(large_tile=None):
    """s2t_gemm_configure: 0 never / 1 automatic / 2, 3 always (where the arguments allow; 256- / 128-row tiles) for the
    LDS-DMA large-tile path of s2t_gemm; None leaves the switch as it is.  Returns the mode in force."""
    return int(L.lib().s2t_gemm_configure(-1 if large_tile is None else int(large_tile)))


def gemm(
    A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int,
    lda: int, ldb: int, ldc: int, a_kmajor=False, b_kmajor=False,
    batch=1, zdiv=1, a_s=(0, 0), b_s=(0, 0), c_s=(0, 0),
    bias: Optional[torch.Tensor] = None, act=None, alpha=1.0,
    residual: Optional[torch.Tensor] = None, ldr=0,
    preact: Optional[torch.Tensor] = None, ldp=0, p_s=(0, 0),
    dact_z: Optional[torch.Tensor] = None, ldz=0, dact=None,
    row_lens: Optional[torch.Tensor] = None, row_T=0, split_k=1, c_atomic=False,
    colsum_a: Optional[torch.Tensor] = None, drop=None, splitk_workspace=True, rows=None,
):
    """Raw s2t_gemm call: C = epilogue(A_op[M,K] @ B_op[K,N]); see include/s2t_hip.h."""
    L.require_cuda(A, B, out, bias, residual, preact, dact_z, row_lens)
    a = L.GemmArgs()
    a.dtype = L.dtype_id(A.dtype)
    assert B.dtype == A.dtype
    a.c_dtype = L.dtype_id(out.dtype)
    a.M, a.N, a.K = M, N, K
    a.a_kmajor, a.b_kmajor = int(a_kmajor), int(b_kmajor)
    a.A, a.lda = A.data_ptr(), lda
    a.B, a.ldb = B.data_ptr(), ldb
    a.C, a.ldc = out.data_ptr(), ldc
    a.batch, a.zdiv = batch, zdiv
    a.a_s0, a.a_s1 = a_s
    a.b_s0, a.b_s1 = b_s
    a.c_s0, a.c_s1 = c_s
    a.bias = _ptr(bias)
    a.bias_dtype = L.dtype_id(bias.dtype) if bias is not None else 0
    a.act = L.ACT_IDS[act]
    a.alpha = alpha
    a.residual, a.ldr = _ptr(residual), ldr
    if residual is not None:
        assert residual.dtype == out.dtype
    a.preact, a.ldp = _ptr(preact), ldp
    a.p_s0, a.p_s1 = p_s
    if preact is not None:
        assert preact.dtype == out.dtype
    a.dact_z, a.ldz, a.dact = _ptr(dact_z), ldz, L.ACT_IDS[dact]
    if dact_z is not None:
        assert dact_z.dtype == out.dtype
    a.row_lens, a.row_T = _mask(row_lens, row_T, rows)
    a.split_k = split_k
    a.c_atomic = int(c_atomic)
    a.colsum_a = _ptr(colsum_a)
    if drop is not None and drop[0] > 0:  # (p, seed tensor (int64 on device), site)
        a.drop_p, a.drop_seed, a.drop_site = float(drop[0]), drop[1].data_ptr(), int(drop[2])
    if colsum_a is not None:
        assert colsum_a.dtype == torch.float32 and a_kmajor
    if split_k > 1 and splitk_workspace:
        need = L.lib().s2t_gemm_ws_floats(C.byref(a))
        if need > 0:
            ws = _scratch("gemm_splitk", need, out.device)
            a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().s2t_gemm(C.byref(a), L.stream_ptr()), "s2t_gemm")
        e1.record()
        Me = _prof_rows(M, row_lens, rows)
        esz, csz = A.element_size(), out.element_size()
        nb = max(batch, 1) * ((Me * K + N * K) * esz + Me * (N // 2 if act == "glu" else N) * csz * (2 if residual is not None else 1))
        GEMM_PROFILE.append((gemm_symbol(a), 2.0 * Me * N * K * max(batch, 1), e0, e1, (Me, N, K, batch), float(nb)))
        return out
    L.check(L.lib().s2t_gemm(C.byref(a), L.stream_ptr()), "s2t_gemm")
    return out


def ffn_fused_supported(x, F, M=None):
    """The row-block kernel covers the recipes' encoder width: bf16, d = 256, F a multiple of 64 up to 4096."""
    return x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] == 256 and F % 64 == 0 and (x.shape[0] + 64) * F < 2 ** 31


def _ffn_pair_ws(a, M, device):
    """Exchange workspace of the several-workgroups-per-block forms (s2t_ffn_pair_ws_bytes): zero at first use, one per stream."""
    need = L.lib().s2t_ffn_pair_ws_bytes(M)
    # one buffer for every row count (the flags sit at its start, the same place for every shape, zero between launches); it
    # only grows, and a new one starts zeroed
    key = ("ffn_pair", str(device), L.stream_ptr())
    t = _WS.get(key)
    if t is None or t.numel() * 4 < need:
        _retire(t)
        t = torch.zeros((need + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = t
    a.pair_ws, a.pair_ws_bytes = t.data_ptr(), t.numel() * 4
    return t


def ffn_configure(pc_mask=None, split=None, fault=None):
    """s2t_ffn_configure: switch the fused feed-forward flavours inside one process (None leaves a switch as it is); returns
    the settings in force as (pc_mask, split_force, fault)."""
    r = L.lib().s2t_ffn_configure(-1 if pc_mask is None else int(pc_mask), -1 if split is None else int(split),
                                  -1 if fault is None else int(fault))
    return r & 7, (r >> 4) & 0xFF, (r >> 12) & 1


def ffn_cu_budget(cus=None):
    """s2t_ffn_cu_budget: compute units the fused feed-forward split may count on (None: query; 0: the device's count)."""
    return int(L.lib().s2t_ffn_cu_budget(-1 if cus is None else int(cus)))


def occupy_cus(n, ms, stop=None, arrived=None, stream=None):
    """s2t_occupy_cus on ``stream`` (a torch.cuda.Stream; default: the current one): n compute units held for at most ms milliseconds."""
    st = stream.cuda_stream if stream is not None else L.stream_ptr()
    L.check(L.lib().s2t_occupy_cus(int(n), int(ms), _ptr(stop), _ptr(arrived), st), "s2t_occupy_cus")


_XCHK = {"pending": []}


def _ffn_exchange_words():
    off = L.lib().s2t_ffn_exchange_error_offset() // 4
    return [(k, t, off) for k, t in _WS.items() if isinstance(k, tuple) and k and k[0] == "ffn_pair"]


def _ffn_exchange_fail(count, ws):
    nb = L.lib().s2t_ffn_exchange_flag_bytes() // 4
    for t in ws:
        t[:nb].zero_()  # a late partner's flag may still be raised: the next launch must start from zero flags
    raise RuntimeError("s2t_amd: %d fused feed-forward launch(es) timed out waiting for a partner workgroup's partial rows "
                       "(csrc/ffn_pc.hip exchange): the results of that step are invalid.  The flags were reset; set "
                       "S2T_FFN_PC_SPLIT=1 to run one workgroup per row block while other kernels share the GPU." % count)


def ffn_exchange_check():
    """Synchronous health check of the fused feed-forward exchange (every workspace of this process): raises RuntimeError
    when a launch since the last check gave up waiting for its partner (include/s2t_hip.h, s2t_ffn_exchange_error_offset)."""
    _XCHK["pending"].clear()
    bad, ws = 0, []
    for _, t, off in _ffn_exchange_words():
        n = int(t.view(torch.int32)[off].item())
        if n:
            bad += n
            ws.append(t)
    if bad:
        _ffn_exchange_fail(bad, ws)


def ffn_exchange_poll():
    """The same check without stalling the host: queues a copy of the error word(s) into pinned memory on the current stream
    and examines the copies queued by EARLIER calls that have completed (the bundled Trainer calls this once per update, so
    a time-out surfaces one update later at the latest; ``ffn_exchange_check`` is the synchronous form)."""
    bad, ws, still = 0, [], []
    for host, ev, t in _XCHK["pending"]:
        if ev.query():
            if int(host[0]):
                bad += int(host[0])
                ws.append(t)
        else:
            still.append((host, ev, t))
    _XCHK["pending"] = still
    if bad:
        _XCHK["pending"] = []
        _ffn_exchange_fail(bad, ws)
    if len(_XCHK["pending"]) < 8:
        for _, t, off in _ffn_exchange_words():
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            host.copy_(t.view(torch.int32)[off:off + 1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _XCHK["pending"].append((host, ev, t))


def ffn_z_rows(M):
    """Rows to allocate for a z buffer that fits either layout (s2t_ffn_z_elems: row blocks of 128)."""
    return (M + 127) // 128 * 128


def ffn_fused_fwd(x, w1, b1, w2, b2, y, *, act, alpha=1.0, residual=None, ln=None, ln_eps=1e-5, end_ln=None, y_ln=None,
                  end_stats=None, end_lens=None, end_T=0, x_ln=None, ln_stats=None, z=None, h=None, drop_h=None,
                  drop_o=None, z_tiled_ok=False, rows=None):
    """s2t_ffn_fused_fwd (include/s2t_hip.h): ``ln`` / ``end_ln`` = (gamma, beta) fp32 of the LayerNorm in front of /
    behind the block; ``ln_stats`` / ``end_stats`` = (mean, rstd) outputs; drops = (p, seed tensor, site) or None.
    ``z_tiled_ok``: z has ``ffn_z_rows(M)`` rows and may be written in the tiled layout; returns True when it was."""
    L.require_cuda(x, w1, w2, y, y_ln, residual, x_ln, z, h)
    M, d = x.shape
    F = w1.shape[0]
    a = L.FfnArgs()
    a.x, a.d, a.M, a.F = x.data_ptr(), d, M, F
    a.ln_gamma, a.ln_beta = (ln[0].data_ptr(), ln[1].data_ptr()) if ln is not None else (None, None)
    a.ln_eps = ln_eps
    a.w1, a.b1, a.w2, a.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    assert w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and b1.dtype == torch.float32
    a.residual, a.y = _ptr(residual), _ptr(y)
    if end_ln is not None:
        a.eln_gamma, a.eln_beta, a.y_ln = end_ln[0].data_ptr(), end_ln[1].data_ptr(), y_ln.data_ptr()
        if end_stats is not None:
            a.eln_mean, a.eln_rstd = end_stats[0].data_ptr(), end_stats[1].data_ptr()
    # (a packed batch's row map also travels without a trailing LayerNorm: it bounds the launch to the live rows)
    a.eln_lens, a.eln_T = _mask(end_lens if end_ln is not None else None, end_T, rows if rows is not None else end_lens)
    a.x_ln = _ptr(x_ln)
    if ln_stats is not None:
        a.ln_mean, a.ln_rstd = ln_stats[0].data_ptr(), ln_stats[1].data_ptr()
    a.z, a.h = _ptr(z), _ptr(h)
    a.act, a.alpha = L.ACT_IDS[act], alpha
    seed = None
    if drop_h is not None and drop_h[0] > 0:
        a.drop_h_p, a.drop_h_site, seed = float(drop_h[0]), int(drop_h[2]), drop_h[1]
    if drop_o is not None and drop_o[0] > 0:
        assert seed is None or seed.data_ptr() == drop_o[1].data_ptr()
        a.drop_o_p, a.drop_o_site, seed = float(drop_o[0]), int(drop_o[2]), drop_o[1]
    a.drop_seed = _ptr(seed)
    _ffn_pair_ws(a, M, x.device)
    if z_tiled_ok and z is not None:
        assert z.numel() >= L.lib().s2t_ffn_z_elems(M, F)
        a.z_tiled_ok = 1
    tiled = bool(L.lib().s2t_ffn_z_tiled(C.byref(a)))
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().s2t_ffn_fused_fwd(C.byref(a), L.stream_ptr()), "s2t_ffn_fused_fwd")
        e1.record()
        buf = C.create_string_buffer(128)
        L.check(L.lib().s2t_ffn_fused_describe(C.byref(a), buf, 128), "s2t_ffn_fused_describe")
        Me = _prof_rows(M, end_lens, rows)
        # algorithmic bytes: every row tensor the call reads or writes once (x doubles as the residual), the two weights
        n256 = 1 + sum(t is not None for t in (y, y_ln, x_ln)) + (1 if residual is not None and residual.data_ptr() != x.data_ptr() else 0)
        nb = Me * 2 * (n256 * d + F * sum(t is not None for t in (z, h))) + 2 * F * d * 2
        GEMM_PROFILE.append((buf.value.decode(), 4.0 * Me * F * d, e0, e1, (Me, F, d, 1), float(nb)))  # the name rocprofv3 prints for this launch
        return tiled
    L.check(L.lib().s2t_ffn_fused_fwd(C.byref(a), L.stream_ptr()), "s2t_ffn_fused_fwd")
    return tiled


def ffn_fused_bwd(dy, w2t, w1t, z, dz, dxn, *, act, alpha=1.0, drop_h=None, ln=None, end=None, z_tiled=False, rows=None):
    """s2t_ffn_fused_bwd (include/s2t_hip.h): dz = alpha * drop_h((dy W2) * act'(z)), dxn = dz W1, from the transposed
    weight copies ``w2t`` [F, 256] and ``w1t`` [256, F].  ``ln`` = dict(x, gamma, mean, rstd, ws, dx[, dres, dx_drop, drop])
    adds the backward of the block's leading LayerNorm (``dxn`` may then be None)."""
    L.require_cuda(dy, w2t, w1t, z, dz, dxn)
    M, d = dy.shape
    F = w2t.shape[0]
    assert w2t.shape == (F, d) and w1t.shape == (d, F) and dz.shape == (M, F)
    assert (z.shape == (M, F)) if not z_tiled else (z.numel() >= L.lib().s2t_ffn_z_elems(M, F))
    assert dxn is None or dxn.shape == (M, d)
    assert all(t is None or (t.dtype == torch.bfloat16 and t.is_contiguous()) for t in (dy, w2t, w1t, z, dz, dxn))
    a = L.FfnBwdArgs()
    a.dy, a.w2t, a.w1t, a.z, a.dz, a.dxn = (_ptr(t) for t in (dy, w2t, w1t, z, dz, dxn))
    a.d, a.M, a.F, a.act, a.alpha = d, M, F, L.ACT_IDS[act], alpha
    a.z_tiled = int(bool(z_tiled))
    if drop_h is not None and drop_h[0] > 0:
        a.drop_h_p, a.drop_h_site, a.drop_seed = float(drop_h[0]), int(drop_h[2]), drop_h[1].data_ptr()
    if ln is not None:
        x, dx = ln["x"], ln["dx"]
        assert x.shape == (M, d) and dx.shape == (M, d) and x.dtype == torch.bfloat16 and x.is_contiguous() and dx.is_contiguous()
        assert ln["ws"].numel() >= LN_REPLICAS * 2 * d and ln["gamma"].dtype == torch.float32
        a.ln_x, a.ln_gamma, a.ln_mean, a.ln_rstd = x.data_ptr(), ln["gamma"].data_ptr(), ln["mean"].data_ptr(), ln["rstd"].data_ptr()
        a.dres, a.ln_ws, a.ln_replicas, a.dx = _ptr(ln.get("dres")), ln["ws"].data_ptr(), LN_REPLICAS, dx.data_ptr()
        if ln.get("dx_drop") is not None:
            dr = ln["drop"]
            assert a.drop_seed is None or a.drop_seed == dr[1].data_ptr()
            a.dx_drop, a.up_drop_p, a.up_drop_site, a.drop_seed = ln["dx_drop"].data_ptr(), float(dr[0]), int(dr[2]), dr[1].data_ptr()
    if end is not None:  # dict(y, gamma, mean, rstd, ws, dres[, lens, T, dy, drop]): trailing LayerNorm's backward in front
        a.end_y, a.end_gamma, a.end_mean, a.end_rstd = (end[k].data_ptr() for k in ("y", "gamma", "mean", "rstd"))
        a.end_lens, a.end_T = _mask(end.get("lens"), int(end.get("T") or 0), rows)
        a.end_ws, a.end_replicas, a.dres_out = end["ws"].data_ptr(), LN_REPLICAS, end["dres"].data_ptr()
        if end.get("dy") is not None:
            dr = end["drop"]
            assert a.drop_seed is None or a.drop_seed == dr[1].data_ptr()
            a.dy_out, a.drop_o_p, a.drop_o_site, a.drop_seed = end["dy"].data_ptr(), float(dr[0]), int(dr[2]), dr[1].data_ptr()
    if end is None:
        a.end_lens, a.end_T = _mask(None, 0, rows)  # packed batch: the live-row bound
    _ffn_pair_ws(a, M, dy.device)
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().s2t_ffn_fused_bwd(C.byref(a), L.stream_ptr()), "s2t_ffn_fused_bwd")
        e1.record()
        buf = C.create_string_buffer(128)
        L.check(L.lib().s2t_ffn_fused_bwd_describe(C.byref(a), buf, 128), "s2t_ffn_fused_bwd_describe")
        Me = _prof_rows(M, end.get("lens") if end is not None else None, rows)
        n256 = 1 + (dxn is not None)  # dy, dxn
        if ln is not None:
            n256 += 2 + (ln.get("dres") is not None) + (ln.get("dx_drop") is not None)  # x, dx, dres, dx_drop
        if end is not None:
            n256 += 2 + (end.get("dy") is not None)  # y, dres_out, dy_out
        nb = Me * 2 * (n256 * d + 2 * F) + 2 * F * d * 2  # + z read, dz written, the two transposed weights
        GEMM_PROFILE.append((buf.value.decode(), 4.0 * Me * F * d, e0, e1, (Me, F, d, 1), float(nb)))
        return
    L.check(L.lib().s2t_ffn_fused_bwd(C.byref(a), L.stream_ptr()), "s2t_ffn_fused_bwd")


def rowblock_dgrad(dy, wt, *, dxn=None, ln=None, rows=None):
    """s2t_rowblock_dgrad (include/s2t_hip.h): dxn = dy @ W from the transposed weight ``wt`` [256, K]; ``ln`` = dict(x, gamma,
    mean, rstd, ws, dx[, dres, lens, T, dx_drop, drop]) adds the backward of the LayerNorm in front of the projection."""
    L.require_cuda(dy, wt, dxn)
    M, Kd = dy.shape
    assert wt.shape == (256, Kd) and dy.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16
    assert dy.is_contiguous() and wt.is_contiguous()
    a = L.RowblockDgradArgs()
    a.dy, a.wt, a.d, a.M, a.K, a.dxn = dy.data_ptr(), wt.data_ptr(), 256, M, Kd, _ptr(dxn)
    if ln is not None:
        x, dx = ln["x"], ln["dx"]
        assert x.shape == (M, 256) and dx.shape == (M, 256) and x.is_contiguous() and dx.is_contiguous()
        assert ln["ws"].numel() >= LN_REPLICAS * 2 * 256
        a.ln_x, a.ln_gamma, a.ln_mean, a.ln_rstd = x.data_ptr(), ln["gamma"].data_ptr(), ln["mean"].data_ptr(), ln["rstd"].data_ptr()
        a.ln_lens, a.ln_T = _mask(ln.get("lens"), int(ln.get("T") or 0), rows)
        a.dres, a.ln_ws, a.ln_replicas, a.dx = _ptr(ln.get("dres")), ln["ws"].data_ptr(), LN_REPLICAS, dx.data_ptr()
        if ln.get("dx_drop") is not None:
            dr = ln["drop"]
            a.dx_drop, a.up_drop_p, a.up_drop_site, a.drop_seed = ln["dx_drop"].data_ptr(), float(dr[0]), int(dr[2]), dr[1].data_ptr()
    else:
        a.ln_lens, a.ln_T = _mask(None, 0, rows)  # packed batch: the live-row bound
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().s2t_rowblock_dgrad(C.byref(a), L.stream_ptr()), "s2t_rowblock_dgrad")
        e1.record()
        Me = _prof_rows(M, ln.get("lens") if ln is not None else None, rows)
        n256 = 1 if ln is None else 2 + (ln.get("dres") is not None) + (ln.get("dx_drop") is not None)  # dxn | x, dx, dres, dx_drop
        nb = Me * 2 * (Kd + 256 * n256) + 256 * Kd * 2
        GEMM_PROFILE.append(("rowblock_dgrad_kernel", 2.0 * Me * 256 * Kd, e0, e1, (Me, 256, Kd, 1), float(nb)))
        return
    L.check(L.lib().s2t_rowblock_dgrad(C.byref(a), L.stream_ptr()), "s2t_rowblock_dgrad")


def transpose_tiles(shapes):
    """int32 [n_tiles, 3] tile list (matrix, row tile, column tile) for s2t_transpose_bf16_batched; shapes = [(rows, cols)]."""
    import numpy as np

    out = []
    for i, (r, c) in enumerate(shapes):
        tr, tc = (r + 63) // 64, (c + 63) // 64
        t = np.empty((tr, tc, 3), dtype=np.int32)
        t[..., 0] = i
        t[..., 1] = np.arange(tr, dtype=np.int32)[:, None]
        t[..., 2] = np.arange(tc, dtype=np.int32)[None, :]
        out.append(t.reshape(-1, 3))
    return np.concatenate(out)


def transpose_batched(table, n, tiles, n_tiles):
    """s2t_transpose_bf16_batched: ``table`` = device uint8 tensor holding n s2t_transpose_item records (24 bytes each),
    ``tiles`` = device int32 tile list (transpose_tiles)."""
    _call("s2t_transpose_bf16_batched", table.data_ptr(), n, tiles.data_ptr(), n_tiles)


RB_CHAIN = os.environ.get("S2T_RB_CHAIN", "1") != "0"  # attention output projection + conv_norm + pointwise conv 1 in one launch


def rowblock_chain(first, second):
    """s2t_rowblock_chain: ``first`` / ``second`` are the keyword dictionaries of two rowblock_gemm calls (``x``, ``w``, ``out``
    as keys too) on the same rows, the second reading the first's output through a LayerNorm; one launch, same results."""
    fa = dict(first)
    a = rowblock_gemm(fa.pop("x"), fa.pop("w"), fa.pop("out"), _args_only=True, **fa)
    sb = dict(second)
    b = rowblock_gemm(sb.pop("x"), sb.pop("w"), sb.pop("out"), _args_only=True, **sb)
    if GEMM_PROFILE is not None:  # (the instrumented step of bench.py times kernels one by one: two launches there)
        fa, sb = dict(first), dict(second)
        rowblock_gemm(fa.pop("x"), fa.pop("w"), fa.pop("out"), **fa)
        rowblock_gemm(sb.pop("x"), sb.pop("w"), sb.pop("out"), **sb)
        return
    L.check(L.lib().s2t_rowblock_chain(C.byref(a), C.byref(b), L.stream_ptr()), "s2t_rowblock_chain")


def rowblock_supported(x, N, act=None):
    """s2t_rowblock_gemm covers the encoder width of the recipes: bf16, K = d = 256, N % 8 == 0 (GLU: N % 64 == 0), N <= 4096
    (the kernel stages the bias row in LDS)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
            and (N % 64 == 0 if act == "glu" else N % 8 == 0) and N <= 4096)


def rowblock_gemm(x, w, out, *, N, ldc, bias=None, act=None, alpha=1.0, residual=None, ldr=0, preact=None, ldp=0, ln=None,
                  ln_eps=1e-5, ln_lens=None, ln_T=0, x_ln=None, ln_stats=None, row_lens=None, row_T=0, drop=None, pre=None, conv=None,
                  rows=None, _args_only=False):
    """s2t_rowblock_gemm (include/s2t_hip.h): out = epilogue(LN(x) @ w[:N]^T); ``ln`` = (gamma, beta) or None;
    ``pre`` = (scale, shift, act): a per-column affine + activation in place of the LayerNorm (masked by ln_lens / ln_T);
    ``conv`` = (taps fp32 [256, 15], frames per utterance[, running_mean, running_var, eps]): the 15-tap depthwise
    convolution over time in front of ``pre`` (with the statistics: ``pre`` holds gamma / beta and the kernel folds them)."""
    L.require_cuda(x, w, out, residual, preact, x_ln)
    M, d = x.shape
    a = L.RowblockArgs()
    a.x, a.d, a.M, a.N = x.data_ptr(), d, M, N
    a.ln_eps = ln_eps
    if ln is not None:
        a.ln_gamma, a.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        a.ln_lens, a.ln_T = _mask(ln_lens, ln_T)
        a.x_ln = _ptr(x_ln)
        if ln_stats is not None:
            a.ln_mean, a.ln_rstd = ln_stats[0].data_ptr(), ln_stats[1].data_ptr()
    elif pre is not None:
        assert pre[0].dtype == torch.float32 and pre[1].dtype == torch.float32
        a.pre_scale, a.pre_shift, a.pre_act = pre[0].data_ptr(), pre[1].data_ptr(), L.ACT_IDS[pre[2]]
        a.ln_lens, a.ln_T = _mask(ln_lens, ln_T)
        a.x_ln = _ptr(x_ln)
        if conv is not None:
            assert conv[0].dtype == torch.float32 and conv[0].is_contiguous() and tuple(conv[0].shape) == (256, 15)
            # (packed batch: the utterance of every window row comes from the row map)
            a.conv_w, a.conv_T = conv[0].data_ptr(), ROWS_PACKED if rows_geom(ln_lens) is not None else int(conv[1])
            if len(conv) > 2:  # (running_mean, running_var, eps): pre[0] / pre[1] are the BatchNorm's gamma / beta
                a.bn_mean, a.bn_var, a.bn_eps = conv[2].data_ptr(), conv[3].data_ptr(), float(conv[4])
    assert w.dtype == torch.bfloat16 and (bias is None or bias.dtype == torch.float32)
    a.w, a.bias = w.data_ptr(), _ptr(bias)
    a.act = L.ACT_IDS[act]
    a.preact, a.ldp = _ptr(preact), ldp
    a.out, a.ldc = out.data_ptr(), ldc
    a.alpha = alpha
    a.row_lens, a.row_T = _mask(row_lens, row_T, rows)
    a.residual, a.ldr = _ptr(residual), ldr
    if drop is not None and drop[0] > 0:
        a.drop_p, a.drop_seed, a.drop_site = float(drop[0]), drop[1].data_ptr(), int(drop[2])
    if _args_only:
        return a
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().s2t_rowblock_gemm(C.byref(a), L.stream_ptr()), "s2t_rowblock_gemm")
        e1.record()
        Me = _prof_rows(M, ln_lens, row_lens, rows)
        nout = N // 2 if act == "glu" else N
        nb = Me * 2 * (d + nout * (2 if residual is not None else 1) + (N if preact is not None else 0) + (d if x_ln is not None else 0)) + N * d * 2
        GEMM_PROFILE.append(("rowblock_gemm_kernel", 2.0 * Me * N * d, e0, e1, (Me, N, d, 1), float(nb)))
        return out
    L.check(L.lib().s2t_rowblock_gemm(C.byref(a), L.stream_ptr()), "s2t_rowblock_gemm")
    return out


# ----------------------------------------------------------------------------------------------
# flat-argument entry points
# ----------------------------------------------------------------------------------------------
def _call(name, *args):
    L.check(getattr(L.lib(), name)(*args, L.stream_ptr()), name)


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps=1e-5, row_lens=None, row_T=0, bound=None):
    """``bound``: lengths tensor of a packed batch when there is no mask (only the live rows are normalised)."""
    L.require_cuda(x, y)
    lp, lt = _mask(row_lens, row_T, bound)
    _call("s2t_layernorm_fwd", L.dtype_id(x.dtype), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
          _ptr(mean), _ptr(rstd), rows, cols, eps, lp, lt)


LN_REPLICAS = 32


_WS = {}


def _scratch(tag, n, device):
    """Uninitialised fp32 scratch that only grows (split-K partial tiles); one buffer per tag and device."""
    key = (tag, str(device), L.stream_ptr())  # per stream: concurrent streams must not share scratch
    t = _WS.get(key)
    if t is None or t.numel() < n:
        _retire(t)
        t = torch.empty(n, dtype=torch.float32, device=device)
        _WS[key] = t
    return t


# Workspaces that only grow are REPLACED when a larger shape comes along.  A captured hipGraph has the old buffer's address baked
# in: once any step of this process has been captured (graphs_hold_workspaces, called by Trainer.capture) a replaced buffer is kept
# alive instead of going back to the allocator — a later replay of that graph must not write into memory that now belongs to
# somebody else (or to nobody: an emptied cache unmaps it).
_WS_RETIRED = []
_WS_GRAPHS = {"captured": False}


def graphs_hold_workspaces():
    _WS_GRAPHS["captured"] = True


def _retire(t):
    if t is not None and _WS_GRAPHS["captured"]:
        _WS_RETIRED.append(t)


def _workspace(tag, n, device):
    """Zero-initialised fp32 scratch reused by every call on the stream: the kernels leave it zeroed on exit."""
    key = (tag, n, str(device))
    t = _WS.get(key)
    if t is None:
        t = torch.zeros(n, dtype=torch.float32, device=device)
        _WS[key] = t
    return t


def layernorm_bwd(x, gamma, dy, mean, rstd, dx, dgamma, dbeta, rows, cols, row_lens=None, row_T=0, dres=None, ws=None,
                  dx_drop=None, drop=None, bound=None):
    """``ws`` given and ``dgamma is None``: leave the partial sums in ``ws`` for layernorm_fold.
    ``dx_drop`` + ``drop`` = (p, seed tensor, site): also write dropout(dx) under that mask (bf16, cols == 256)."""
    if ws is None:
        ws = _workspace("ln", LN_REPLICAS * 2 * cols, x.device)
    p, seed, site = (float(drop[0]), drop[1].data_ptr(), int(drop[2])) if dx_drop is not None else (0.0, None, 0)
    _call("s2t_layernorm_bwd", L.dtype_id(x.dtype), x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), mean.data_ptr(),
          rstd.data_ptr(), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), ws.data_ptr(), LN_REPLICAS, rows, cols,
          *_mask(row_lens, row_T, bound), _ptr(dres), _ptr(dx_drop), p, seed, site)


class _LnFoldEntry(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("cols", C.c_int32), ("reserved", C.c_int32)]


def layernorm_fold(entries):
    """entries: list of (ws, dgamma, dbeta, cols) tensors/ints."""
    n = len(entries)
    if n == 0:
        return
    arr = (_LnFoldEntry * n)()
    for i, (ws, dg, db, cols) in enumerate(entries):
        arr[i].ws, arr[i].dgamma, arr[i].dbeta, arr[i].cols = ws.data_ptr(), dg.data_ptr(), db.data_ptr(), cols
    _call("s2t_layernorm_fold", C.cast(arr, C.c_void_p), n, LN_REPLICAS)


def _drop3(drop):
    if drop is None or drop[0] <= 0:
        return 0.0, None, 0
    return float(drop[0]), drop[1].data_ptr(), int(drop[2])


def attn_softmax_fwd(S, ldS, BD, ldBD, P, ldP, Z, H, Tq, Tk, scale, key_lens=None, causal=False, clamp=False,
                     Pdrop=None, drop=None):
    assert S.dtype == torch.float32 and (BD is None or BD.dtype == torch.float32)
    dp, ds, dsite = _drop3(drop)
    _call("s2t_attn_softmax_fwd", L.dtype_id(P.dtype), S.data_ptr(), ldS, _ptr(BD), ldBD, P.data_ptr(), ldP, Z, H, Tq,
          Tk, scale, _ptr(key_lens), int(causal), int(clamp), _ptr(Pdrop), dp, ds, dsite)


def attn_softmax_bwd(P, ldP, dP, ldDP, dS, ldDS, dBD, ldDBD, Z, H, Tq, Tk, scale, drop=None):
    assert dP.dtype == torch.float32
    dp, ds, dsite = _drop3(drop)
    _call("s2t_attn_softmax_bwd", L.dtype_id(P.dtype), P.data_ptr(), ldP, dP.data_ptr(), ldDP, dS.data_ptr(), ldDS,
          _ptr(dBD), ldDBD, Z, H, Tq, Tk, scale, dp, ds, dsite)


def dropout(x, ldx, out, ldo, rows, cols, drop):
    dp, ds, dsite = _drop3(drop)
    _call("s2t_dropout", L.dtype_id(x.dtype), x.data_ptr(), ldx, out.data_ptr(), ldo, rows, cols, dp, ds, dsite)


def bias_add_rows(x, ldx, bias, out, ldo, rows, n):
    _call("s2t_bias_add_rows", L.dtype_id(x.dtype), x.data_ptr(), ldx, bias.data_ptr(), out.data_ptr(), ldo, rows, n)


def add_positions(x, tab, lens, rows, T, d, scale=1.0, pos_offset=2):
    lp, lt = _mask(lens, T)
    _call("s2t_add_positions", L.dtype_id(x.dtype), x.data_ptr(), _ptr(tab), lp, rows, lt if lp is not None else T, d, scale,
          pos_offset)


def mask_rows(x, lens, rows, T, d):
    lp, lt = _mask(lens, T)
    _call("s2t_mask_rows", L.dtype_id(x.dtype), x.data_ptr(), lp, rows, lt, d)


def embedding_fwd(tokens, pos, E, tab, out, n, d, scale):
    _call("s2t_embedding_fwd", L.dtype_id(E.dtype), tokens.data_ptr(), _ptr(pos), E.data_ptr(), _ptr(tab), out.data_ptr(),
          n, d, scale)


def embedding_bwd(tokens, dout, dE, n, d, scale, pad_idx):
    _call("s2t_embedding_bwd", L.dtype_id(dout.dtype), tokens.data_ptr(), dout.data_ptr(), dE.data_ptr(), n, d, scale,
          pad_idx)


def glu_bwd(Z, dY, dZ, rows, n, lens=None, T=0, out_pad=0):
    """``out_pad``: dZ is laid out with ``out_pad`` extra rows behind every T rows (those rows are not written)."""
    _call("s2t_glu_bwd", L.dtype_id(Z.dtype), Z.data_ptr(), dY.data_ptr(), dZ.data_ptr(), rows, n, _ptr(lens), T, out_pad)


def colsum_accum(dY, ld, db, rows, n):
    _call("s2t_colsum_accum", L.dtype_id(dY.dtype), dY.data_ptr(), ld, db.data_ptr(), rows, n)


def cast_bf16_to_f32(src, dst, n, scale=1.0):
    _call("s2t_cast_bf16_to_f32", src.data_ptr(), dst.data_ptr(), n, float(scale))


def cast_f32_to_bf16(src, dst, n):
    _call("s2t_cast_f32_to_bf16", src.data_ptr(), dst.data_ptr(), n)


def axpy(a, b, y, alpha, n):
    _call("s2t_axpy", L.dtype_id(a.dtype), a.data_ptr(), b.data_ptr(), y.data_ptr(), alpha, n)


def adam_step(p, g, m, v, shadow, n, beta1, beta2, eps, wd, hyper):
    _call("s2t_adam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(shadow), n, beta1, beta2, eps,
          wd, hyper.data_ptr())


def clip_coef(sumsq, max_norm, mult, hyper):
    _call("s2t_clip_coef", sumsq.data_ptr(), max_norm, mult, hyper.data_ptr())


def sumsq_accum(g, n, out):
    _call("s2t_sumsq_accum", g.data_ptr(), n, out.data_ptr())


def dwconv_fwd(x, w, y, B, T, C, K, flip=False, scale=None, shift=None, act=None, lens=None, stats=None):
    _call("s2t_dwconv_fwd", L.dtype_id(x.dtype), x.data_ptr(), w.data_ptr(), y.data_ptr(), B, T, C, K, int(flip),
          _ptr(scale), _ptr(shift), L.ACT_IDS[act], _ptr(lens), _cu(lens), _ptr(stats))


def dwconv_bn_eval_fwd(x, w, y, B, T, C, K, gamma, beta, running_mean, running_var, eps, act, lens=None):
    _call("s2t_dwconv_bn_eval_fwd", L.dtype_id(x.dtype), x.data_ptr(), w.data_ptr(), y.data_ptr(), B, T, C, K, gamma.data_ptr(),
          beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), float(eps), L.ACT_IDS[act], _ptr(lens), _cu(lens))


def dwconv_bwd_weight(G, dD, dw, B, T, C, K):
    rows = L.lib().s2t_dwconv_wgrad_partials(B, T)  # one plain-stored partial row per workgroup, folded in fixed order
    ws = _scratch("dw", rows * C * K, G.device)
    _call("s2t_dwconv_bwd_weight", L.dtype_id(G.dtype), G.data_ptr(), dD.data_ptr(), dw.data_ptr(), ws.data_ptr(),
          rows, B, T, C, K)


def conv_bwd_fused(D, dA, G, Z, w, scale, shift, mean, rstd, sums, count, act, lens, dZ, dw, B, T, C, Kw, defer_slot=None):
    """s2t_conv_bwd_fused (include/s2t_hip.h); bf16 only.  ``defer_slot`` = k: the depthwise weight-gradient partial rows stay in
    the k-th scratch block, returned with their row count for a later ``rows_fold_add`` over several modules."""
    L.require_cuda(D, dA, G, Z, dZ)
    assert all(t.dtype == torch.bfloat16 and t.is_contiguous() for t in (D, dA, G, Z, dZ))
    rows = B * ((T + 31) // 32)
    ws = _scratch("dw_fused" if defer_slot is None else "dw_fused%d" % defer_slot, rows * C * Kw, D.device)
    _call("s2t_conv_bwd_fused", D.data_ptr(), dA.data_ptr(), G.data_ptr(), Z.data_ptr(), w.data_ptr(), scale.data_ptr(),
          shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(), float(count), L.ACT_IDS[act], _ptr(lens),
          _cu(lens), dZ.data_ptr(), dw.data_ptr() if defer_slot is None else None, ws.data_ptr(), B, T, C, Kw)
    return ws, rows


def dwconv_stat_partials(B, T):
    return L.lib().s2t_dwconv_stat_partials(B, T)


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd, C):
    """``stats``: [partials][2][C] rows written by dwconv_fwd (training) or None."""
    _call("s2t_bn_finalize", _ptr(stats), stats.shape[0] if stats is not None else 0, float(count), gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean),
          _ptr(running_var), momentum, eps, int(training), scale.data_ptr(), shift.data_ptr(), _ptr(mean), _ptr(rstd), C)


def bn_act_fwd(D, out, scale, shift, act, rows, C, lens=None, T=0):
    _call("s2t_bn_act_fwd", L.dtype_id(D.dtype), D.data_ptr(), out.data_ptr(), scale.data_ptr(), shift.data_ptr(),
          L.ACT_IDS[act], rows, C, *_mask(lens, T))


def bn_act_bwd(D, dOut, dD, scale, shift, mean, rstd, sums, count, act, rows, C, lens=None, T=0, dgamma=None, dbeta=None):
    """``dgamma`` / ``dbeta`` (fp32 [C], optional): the folded sums are also ADDED to them (the parameter gradients)."""
    ws = _scratch("bn_bwd", L.lib().s2t_bn_bwd_partials(rows) * 2 * C, D.device)
    _call("s2t_bn_act_bwd", L.dtype_id(D.dtype), D.data_ptr(), dOut.data_ptr(), _ptr(dD), scale.data_ptr(),
          shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(), ws.data_ptr(), _ptr(dgamma), _ptr(dbeta),
          float(count), L.ACT_IDS[act], rows, C, *_mask(lens, T))


def argmax_lse(logits, ld, rows, V, idx=None, top_lp=None, lse=None, bound=None):
    _call("s2t_argmax_lse", L.dtype_id(logits.dtype), logits.data_ptr(), ld, rows, V, _ptr(idx), _ptr(top_lp), _ptr(lse),
          _live(bound))


def ctc_head_greedy(x, w, bias, idx=None, top_lp=None, lse=None, bound=None):
    """s2t_ctc_head_greedy (include/s2t_hip.h): arg-max / its log-probability / logsumexp of x @ w^T + bias per row, fp32 arithmetic, the
    logits never stored.  x: bf16 [M, 256] rows (row stride % 8 == 0), w: bf16 [V, 256], bias: fp32 [V] or None."""
    L.require_cuda(x, w, bias, idx, top_lp, lse)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.shape[1] == 256 and w.shape[1] == 256 and x.stride(1) == 1
    assert w.is_contiguous() and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))
    _call("s2t_ctc_head_greedy", x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), x.shape[0], w.shape[0], _ptr(idx), _ptr(top_lp),
          _ptr(lse), _live(bound))


def ctc_head_greedy_supported(x, w):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.stride(1) == 1 and x.stride(0) % 8 == 0
            and w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] == 256 and w.is_contiguous() and w.shape[0] >= 128
            and w.shape[0] * 512 < 2 ** 32)


def add_colsum2(a, lda, b, ldb, du, dv, rows, n):
    _call("s2t_add_colsum2", L.dtype_id(a.dtype), a.data_ptr(), lda, b.data_ptr(), ldb, du.data_ptr(), dv.data_ptr(), rows, n)


def ctc_collapse(idx, top_lp, lens, B, T, blank, out_tokens, out_lens, out_scores, rows=None):
    _call("s2t_ctc_collapse", idx.data_ptr(), top_lp.data_ptr(), lens.data_ptr(), _cu(rows if rows is not None else lens), B, T,
          blank, out_tokens.data_ptr(),
          out_lens.data_ptr(), out_scores.data_ptr())


def ls_cross_entropy(logits, ld, rows, V, target, pad_idx, eps, dlogits, ldd, sums, bound=None):
    ws = _scratch("ls_ce", max(int(rows), 1) * 4, logits.device)
    _call("s2t_ls_cross_entropy", L.dtype_id(logits.dtype), logits.data_ptr(), ld, rows, V, target.data_ptr(), pad_idx,
          eps, _ptr(dlogits), ldd, sums.data_ptr(), ws.data_ptr(), _live(bound))


def ctc_loss_fwd(logits, ld, B, T, V, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll,
                 force_emits=None, paths=None, rows=None):
    _call("s2t_ctc_loss_fwd", L.dtype_id(logits.dtype), logits.data_ptr(), ld, B, T, V, lse.data_ptr(),
          targets.data_ptr(), ldt, tgt_lens.data_ptr(), in_lens.data_ptr(), blank, alpha.data_ptr(), _ptr(beta),
          Lmax, nll.data_ptr(), _ptr(force_emits), _ptr(paths), _cu(rows))


def ctc_loss_bwd(logits, ld, B, T, V, lse, targets, ldt, tgt_lens, in_lens, blank, alpha, beta, Lmax, nll, gscale,
                 grad, ldg, wrt_logprobs=False, gscale_dev=None, rows=None):
    _call("s2t_ctc_loss_bwd", L.dtype_id(logits.dtype), logits.data_ptr(), ld, B, T, V, lse.data_ptr(),
          targets.data_ptr(), ldt, tgt_lens.data_ptr(), in_lens.data_ptr(), blank, alpha.data_ptr(), beta.data_ptr(),
          Lmax, nll.data_ptr(), gscale, _ptr(gscale_dev), grad.data_ptr(), ldg, int(wrt_logprobs), _cu(rows))


def ctc_backtrace(alpha, paths, tgt_lens, in_lens, B, T, Lmax, states):
    _call("s2t_ctc_backtrace", alpha.data_ptr(), paths.data_ptr(), tgt_lens.data_ptr(), in_lens.data_ptr(), B, T, Lmax,
          states.data_ptr())


def ctc_prefix_init(lp, T, in_lens, sent, blank, r0):
    """lp: fp32 [B*T, V] log-probabilities (batch-major rows); r0: [R, T, 2]."""
    _call("s2t_ctc_prefix_init", lp.data_ptr(), lp.stride(0), T, in_lens.data_ptr(), sent.data_ptr(), r0.shape[0], blank,
          r0.data_ptr())


def ctc_prefix_score(lp, T, in_lens, sent, r_prev, last, out_len, cand, blank, eos, psi, r_new=None):
    """cand: int64 [R, K]; psi: fp32 [R, K]; r_new: optional fp32 [R, K, T, 2]."""
    R, Kc = cand.shape
    _call("s2t_ctc_prefix_score", lp.data_ptr(), lp.stride(0), T, in_lens.data_ptr(), sent.data_ptr(), r_prev.data_ptr(),
          last.data_ptr(), out_len, cand.data_ptr(), R, Kc, blank, eos, psi.data_ptr(), _ptr(r_new))


def dwpool_stat_partials(B, Tout):
    return L.lib().s2t_dwpool_stat_partials(B, Tout)


def dwpool_fwd(x, w, bias, y, B, Tin, C, r, stats=None):
    _call("s2t_dwpool_fwd", L.dtype_id(x.dtype), x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Tin, C, r,
          _ptr(stats))


def dwpool_bwd(x, w, dy, dx, dw, db, B, Tin, C, r):
    _call("s2t_dwpool_bwd", L.dtype_id(x.dtype), x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
          db.data_ptr(), B, Tin, C, r)


def ctc_compress_plan(logits, lse, lens, B, T, blank, threshold, src, new_lens):
    _call("s2t_ctc_compress_plan", L.dtype_id(logits.dtype), logits.data_ptr(), logits.stride(0), lse.data_ptr(),
          lens.data_ptr(), B, T, blank, float(threshold), src.data_ptr(), new_lens.data_ptr())


def compress_rows(x, out, src, new_lens, B, T, Tn, C, scatter=False):
    _call("s2t_compress_rows", L.dtype_id(x.dtype), x.data_ptr(), out.data_ptr(), src.data_ptr(), new_lens.data_ptr(), B, T,
          Tn, C, 1 if scatter else 0)


def row_softmax_fwd(x, ldx, p, ldp, rows, V, inv_tau=1.0, bound=None):
    """``bound``: lengths tensor of a packed batch — only its live rows are computed."""
    _call("s2t_row_softmax_fwd", L.dtype_id(x.dtype), x.data_ptr(), ldx, p.data_ptr(), ldp, rows, V, inv_tau, _live(bound))


def row_softmax_bwd(p, ldp, dp, lddp, dx, lddx, rows, V, inv_tau=1.0, bound=None):
    _call("s2t_row_softmax_bwd", L.dtype_id(p.dtype), p.data_ptr(), ldp, dp.data_ptr(), lddp, dx.data_ptr(), lddx, rows, V,
          inv_tau, _live(bound))


def attn_fused_fwd(q, q_sb, q_sr, k, k_sb, k_sr, v, v_sb, v_sr, o, o_sb, o_sr, lse, B, H, Tq, Tk, dk, key_lens, causal, scale,
                   pos_p=None, p_sr=0, pos_u=None, pos_v=None, drop=None, q_rows=None, k_rows=None, o_lo=None):
    """``q_rows`` / ``k_rows``: lengths tensor of a packed batch on the query / key side (rows of utterance b from cu[b]).
    ``o_lo``: optional bf16 buffer of o's layout for the rounding remainder of the output (read back by attn_fused_bwd)."""
    assert q.dtype == torch.bfloat16 and (o_lo is None or (o_lo.dtype == torch.bfloat16 and o_lo.shape == o.shape))
    dp, ds, dsite = _drop3(drop)
    _call("s2t_attn_fused_fwd", q.data_ptr(), q_sb, q_sr, k.data_ptr(), k_sb, k_sr, v.data_ptr(), v_sb, v_sr, o.data_ptr(),
          o_sb, o_sr, _ptr(lse), B, H, Tq, Tk, dk, _ptr(key_lens), int(causal), scale, _ptr(pos_p), p_sr, _ptr(pos_u),
          _ptr(pos_v), dp, ds, dsite, _cu(q_rows), _cu(k_rows), _ptr(o_lo))


def attn_fused_bwd(q, q_sb, q_sr, k, k_sb, k_sr, v, v_sb, v_sr, o, dO, o_sb, o_sr, lse, delta, dq, dk, dv, dbd, ldb, B, H, Tq,
                   Tk, dkd, key_lens, causal, scale, pos_p=None, p_sr=0, pos_u=None, pos_v=None, drop=None, dbd_band_only=False,
                   pos_pt=None, pt_ld=0, dpos_u=None, dpos_v=None, qv_out=None, q_rows=None, k_rows=None, o_lo=None):
    """``pos_pt``: a VIEW starting at position n = 0 of the zero-padded transposed projections (see include/s2t_hip.h)."""
    dp, ds, dsite = _drop3(drop)
    _call("s2t_attn_fused_bwd", q.data_ptr(), q_sb, q_sr, k.data_ptr(), k_sb, k_sr, v.data_ptr(), v_sb, v_sr, o.data_ptr(),
          dO.data_ptr(), o_sb, o_sr, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _ptr(dbd),
          ldb, B, H, Tq, Tk, dkd, _ptr(key_lens), int(causal), scale, _ptr(pos_p), p_sr, _ptr(pos_u), _ptr(pos_v), dp, ds,
          dsite, int(dbd_band_only), _ptr(pos_pt), pt_ld, _ptr(dpos_u), _ptr(dpos_v), _ptr(qv_out), _cu(q_rows), _cu(k_rows),
          _ptr(o_lo))


def fbank(wave, n_samples, feat, max_frames, win, shift, nfft, window, mel_t, preemph=0.97, remove_dc=True,
          log_floor=1.1920928955078125e-07):
    B = wave.shape[0]
    assert wave.dtype == torch.float32 and feat.dtype == torch.float32 and n_samples.dtype == torch.int32
    _call("s2t_fbank", wave.data_ptr(), wave.stride(0), n_samples.data_ptr(), feat.data_ptr(), feat.stride(0), max_frames, B,
          win, shift, nfft, window.data_ptr(), mel_t.data_ptr(), mel_t.shape[1], preemph, int(remove_dc), log_floor)


def utterance_cmvn(x, y, n_frames, norm_means=True, norm_vars=True):
    B, T, Cf = x.shape
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and n_frames.dtype == torch.int32 and x.is_contiguous()
    _call("s2t_utterance_cmvn", x.data_ptr(), y.data_ptr(), n_frames.data_ptr(), T * Cf, B, Cf, int(norm_means),
          int(norm_vars))


def relpos_dqv(dbd, ldb, pos_pt, pt_ld, dq, dq_sb, dq_sr, dpos_u, dpos_v, B, H, Tq, dk, replicas=1, replica_stride=0):
    L.require_cuda(dbd, pos_pt, dq, dpos_u, dpos_v)
    assert dbd.dtype == torch.bfloat16 and pos_pt.dtype == torch.bfloat16 and dq.dtype == torch.bfloat16
    assert dpos_u.dtype == torch.float32 and dpos_v.dtype == torch.float32
    _call("s2t_relpos_dqv", dbd.data_ptr(), ldb, pos_pt.data_ptr(), pt_ld, dq.data_ptr(), dq_sb, dq_sr, dpos_u.data_ptr(),
          dpos_v.data_ptr(), replicas, replica_stride, B, H, Tq, dk)


def relpos_glue(dbd, ldb, pos_p, p_sr, qv, dq, dq_sb, dq_sr, dpos_u, dpos_v, dp, B, H, Tq, dk, replicas=1, replica_stride=0,
                defer_slot=None, rows=None):
    """s2t_relpos_glue: dq += (Q+v) branch, both bias-gradient column sums, and dp (fp32 [2Tq-1, H*dk], overwritten) in one
    pass over dbd.  ``defer_slot`` = k: the per-utterance partial table goes to the k-th scratch table and is returned for a
    later ``relpos_dp_reduce`` over several layers (dp is not written by this call)."""
    L.require_cuda(dbd, pos_p, qv, dq, dpos_u, dpos_v, dp)
    assert dbd.dtype == torch.bfloat16 and pos_p.dtype == torch.bfloat16 and qv.dtype == torch.bfloat16 and dq.dtype == torch.bfloat16
    assert dp.dtype == torch.float32 and dp.is_contiguous() and dp.shape == (2 * Tq - 1, H * dk) and qv.is_contiguous()
    tag = "relpos_dp_part" if defer_slot is None else "relpos_dp_part%d" % defer_slot
    part = _scratch(tag, (B * (2 * Tq - 1) * H * dk + 1) // 2, dbd.device)  # fp32 scratch holding the bf16 partials
    # beyond 256 frames: the remainder of dq's running bf16 sum between position chunks (one buffer for every layer: a call
    # reads only what it wrote itself)
    lo = _scratch("relpos_dq_lo", (B * Tq * H * dk + 1) // 2, dbd.device) if Tq > 256 else None
    _call("s2t_relpos_glue", dbd.data_ptr(), ldb, pos_p.data_ptr(), p_sr, qv.data_ptr(), dq.data_ptr(), dq_sb, dq_sr,
          dpos_u.data_ptr(), dpos_v.data_ptr(), replicas, replica_stride, part.data_ptr(),
          dp.data_ptr() if defer_slot is None else None, B, H, Tq, dk, _cu(rows), _ptr(lo))
    return part


def relpos_attn_bwd(q, q_sb, q_sr, k, k_sb, k_sr, v, v_sb, v_sr, o, dO, o_sb, o_sr, lse, dq, dk, dv, pos_p, p_sr, pos_u, pos_v,
                    dpos_u, dpos_v, B, H, T, dkd, key_lens, scale, drop=None, replicas=1, replica_stride=0, defer_slot=None,
                    rows=None):
    """s2t_relpos_attn_bwd: the relative-position self-attention backward in one launch (T <= 256).  Returns the scratch that
    holds the per-utterance partial tables of the position gradient (``relpos_dp_reduce`` sums them); ``defer_slot`` as
    ``relpos_glue``."""
    L.require_cuda(q, k, v, o, dO, lse, dq, dk, dv, pos_p, pos_u, pos_v, dpos_u, dpos_v)
    assert all(t.dtype == torch.bfloat16 for t in (q, k, v, o, dO, dq, dk, dv, pos_p))
    assert lse.dtype == torch.float32 and dpos_u.dtype == torch.float32 and dpos_v.dtype == torch.float32
    tag = "relpos_dp_part" if defer_slot is None else "relpos_dp_part%d" % defer_slot
    part = _scratch(tag, (B * (2 * T - 1) * H * dkd + 1) // 2, q.device)  # fp32 scratch holding the bf16 partials
    dp, ds, dsite = _drop3(drop)
    _call("s2t_relpos_attn_bwd", q.data_ptr(), q_sb, q_sr, k.data_ptr(), k_sb, k_sr, v.data_ptr(), v_sb, v_sr, o.data_ptr(),
          dO.data_ptr(), o_sb, o_sr, lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), pos_p.data_ptr(), p_sr,
          pos_u.data_ptr(), pos_v.data_ptr(), dpos_u.data_ptr(), dpos_v.data_ptr(), replicas, replica_stride, part.data_ptr(),
          B, H, T, dkd, _ptr(key_lens), scale, dp, ds, dsite, _cu(rows))
    return part


def attn_bwd_one_pass(q, q_sb, q_sr, k, k_sb, k_sr, v, v_sb, v_sr, o, dO, o_sb, o_sr, lse, dq, dk, dv, B, H, Tq, Tk, dkd, key_lens,
                      causal, scale, drop=None, q_rows=None, k_rows=None, o_lo=None):
    """s2t_attn_bwd_one_pass: plain attention backward in one launch (Tk <= 256): dq, dk, dv; no delta buffer."""
    L.require_cuda(q, k, v, o, dO, lse, dq, dk, dv)
    assert all(t.dtype == torch.bfloat16 for t in (q, k, v, o, dO, dq, dk, dv)) and lse.dtype == torch.float32
    dp, ds, dsite = _drop3(drop)
    _call("s2t_attn_bwd_one_pass", q.data_ptr(), q_sb, q_sr, k.data_ptr(), k_sb, k_sr, v.data_ptr(), v_sb, v_sr, o.data_ptr(),
          dO.data_ptr(), o_sb, o_sr, lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Tq, Tk, dkd, _ptr(key_lens),
          int(causal), scale, dp, ds, dsite, _cu(q_rows), _cu(k_rows), _ptr(o_lo))


def _ptr_array(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def relpos_dp_reduce(parts, dps, B, H, Tq, dk):
    """s2t_relpos_dp_reduce: dps[i] = fp32 sum over the B per-utterance tables of parts[i], all i in one launch."""
    L.require_cuda(*parts, *dps)
    _call("s2t_relpos_dp_reduce", _ptr_array(parts), _ptr_array(dps), len(parts), B, H, Tq, dk)


def rows_fold_add(partials, outs, rows, n):
    """s2t_rows_fold_add: outs[i][0:n] += fixed-order sum of the ``rows`` partial rows of partials[i], all i in one launch."""
    L.require_cuda(*partials, *outs)
    assert all(t.dtype == torch.float32 for t in partials) and all(t.dtype == torch.float32 for t in outs)
    _call("s2t_rows_fold_add", _ptr_array(partials), _ptr_array(outs), len(partials), rows, n)


def time_warp(x, y, n_frames, warp, mean_out=None):
    B, T, Cf = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    assert warp.dtype == torch.int32 and warp.shape == (B, 2) and n_frames.dtype == torch.int32
    L.require_cuda(x, y, n_frames, warp, mean_out)
    _call("s2t_time_warp", x.data_ptr(), y.data_ptr(), n_frames.data_ptr(), T * Cf, B, T, Cf, warp.data_ptr(), _ptr(mean_out))


def specaugment(x, n_frames, masks, n_freq, n_time, value, value_is_mean):
    B, T, Cf = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and masks.dtype == torch.int32 and value.dtype == torch.float32
    _call("s2t_specaugment", x.data_ptr(), n_frames.data_ptr(), T * Cf, B, T, Cf, masks.data_ptr(), n_freq, n_time,
          value.data_ptr(), int(value_is_mean))


def wgrad_grouped256(problems, n_problems, items, n_items, tiles, n_tiles, ws):
    _call("s2t_wgrad_grouped256", problems.data_ptr(), n_problems, items.data_ptr(), n_items, tiles.data_ptr(), n_tiles,
          ws.data_ptr())


def wgrad_grouped(problems, n_problems, items, n_items, tiles, n_tiles, ws, any_k_tail):
    _call("s2t_wgrad_grouped", problems.data_ptr(), n_problems, items.data_ptr(), n_items, tiles.data_ptr(), n_tiles,
          ws.data_ptr(), int(any_k_tail))
