"""ctypes binding of libs2t_hip.so (the drop-in boundary, include/s2t_hip.h).

Loading is lazy and LOUD: a missing library raises ``RuntimeError`` — the product path never
falls back to a CPU implementation.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libs2t_hip.so")

S2T_F32, S2T_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_GLU = 0, 1, 2, 3
ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "linear": ACT_NONE, "relu": ACT_RELU, "swish": ACT_SWISH, "glu": ACT_GLU}

_lib = None


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("c_dtype", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("batch", C.c_int32), ("zdiv", C.c_int32),
        ("a_s0", C.c_int64), ("a_s1", C.c_int64), ("b_s0", C.c_int64), ("b_s1", C.c_int64),
        ("c_s0", C.c_int64), ("c_s1", C.c_int64),
        ("bias", C.c_void_p), ("bias_dtype", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("preact", C.c_void_p), ("ldp", C.c_int64), ("p_s0", C.c_int64), ("p_s1", C.c_int64),
        ("dact_z", C.c_void_p), ("ldz", C.c_int64), ("dact", C.c_int32),
        ("row_lens", C.c_void_p), ("row_T", C.c_int32),
        ("split_k", C.c_int32),
    ]


def lib():
    """Return the loaded library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "s2t_amd: %s is missing — run `python -m s2t_amd.build` (or __graft_entry__.build()); "
                "there is no CPU fallback for the HIP hot path" % LIB_PATH
            )
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(l):
    l.s2t_version.restype = C.c_int
    l.s2t_device_cu_count.restype = C.c_int
    l.s2t_gemm.restype = C.c_int
    l.s2t_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    for name, argtypes in _PROTOS.items():
        fn = getattr(l, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes


# name -> argtypes for the flat-argument entry points (filled in by ops modules' declarations below)
_PROTOS = {}


def dtype_id(t: torch.dtype) -> int:
    if t == torch.float32:
        return S2T_F32
    if t == torch.bfloat16:
        return S2T_BF16
    raise TypeError("s2t_amd supports float32 and bfloat16 tensors, got %s" % t)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError("libs2t_hip: %s failed with status %d%s" % (
            what, status, " (argument error)" if status < 0 else " (hipError_t)"))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("s2t_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)
