"""Thin, allocation-explicit Python wrappers over the C-ABI kernels (no autograd here)."""
import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def gemm(
    A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int,
    lda: int, ldb: int, ldc: int, a_kmajor=False, b_kmajor=False,
    batch=1, zdiv=1, a_s=(0, 0), b_s=(0, 0), c_s=(0, 0),
    bias: Optional[torch.Tensor] = None, act=None, alpha=1.0,
    residual: Optional[torch.Tensor] = None, ldr=0,
    preact: Optional[torch.Tensor] = None, ldp=0, p_s=(0, 0),
    dact_z: Optional[torch.Tensor] = None, ldz=0, dact=None,
    row_lens: Optional[torch.Tensor] = None, row_T=0, split_k=1,
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
    a.row_lens, a.row_T = _ptr(row_lens), row_T
    if row_lens is not None:
        assert row_lens.dtype == torch.int32
    a.split_k = split_k
    L.check(L.lib().s2t_gemm(C.byref(a), L.stream_ptr()), "s2t_gemm")
    return out
