"""ctypes binding of libs2t_hip.so (the drop-in boundary declared in include/s2t_hip.h).

The prototypes are parsed from the header itself, so the binding cannot drift from the C-ABI.
Loading is lazy and LOUD: a missing library raises ``RuntimeError`` — the product path never
falls back to a CPU implementation.
"""
import ctypes as C
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# S2T_HIP_LIB points the binding at another build of the same C-ABI (kernel experiments, a system-wide install)
LIB_PATH = os.environ.get("S2T_HIP_LIB") or os.path.join(HERE, "lib", "libs2t_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "s2t_hip.h")

S2T_F32, S2T_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_GLU = 0, 1, 2, 3
ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "linear": ACT_NONE, "relu": ACT_RELU, "swish": ACT_SWISH, "glu": ACT_GLU}

_lib = None


class GemmArgs(C.Structure):
    """Mirror of ``struct s2t_gemm_args`` (include/s2t_hip.h)."""

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
        ("c_atomic", C.c_int32),
        ("colsum_a", C.c_void_p),
        ("drop_p", C.c_float), ("drop_site", C.c_uint32), ("drop_seed", C.c_void_p),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
    ]


class FfnArgs(C.Structure):
    """Mirror of ``struct s2t_ffn_args`` (include/s2t_hip.h)."""

    _fields_ = [
        ("x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("d", C.c_int32),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("residual", C.c_void_p), ("y", C.c_void_p),
        ("eln_gamma", C.c_void_p), ("eln_beta", C.c_void_p), ("y_ln", C.c_void_p),
        ("eln_mean", C.c_void_p), ("eln_rstd", C.c_void_p), ("eln_lens", C.c_void_p), ("eln_T", C.c_int32),
        ("x_ln", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p),
        ("z", C.c_void_p), ("h", C.c_void_p),
        ("M", C.c_int32), ("F", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
        ("drop_h_p", C.c_float), ("drop_h_site", C.c_uint32), ("drop_o_p", C.c_float), ("drop_o_site", C.c_uint32),
        ("drop_seed", C.c_void_p),
        ("pair_ws", C.c_void_p), ("pair_ws_bytes", C.c_int64), ("z_tiled_ok", C.c_int32),
    ]


class FfnBwdArgs(C.Structure):
    """Mirror of ``struct s2t_ffn_bwd_args`` (include/s2t_hip.h)."""

    _fields_ = [
        ("dy", C.c_void_p), ("w2t", C.c_void_p), ("w1t", C.c_void_p), ("z", C.c_void_p), ("dz", C.c_void_p),
        ("dxn", C.c_void_p), ("d", C.c_int32), ("M", C.c_int32), ("F", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float), ("drop_h_p", C.c_float), ("drop_h_site", C.c_uint32), ("drop_seed", C.c_void_p),
        ("ln_x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p),
        ("dres", C.c_void_p), ("ln_ws", C.c_void_p), ("ln_replicas", C.c_int32), ("dx", C.c_void_p),
        ("dx_drop", C.c_void_p), ("up_drop_p", C.c_float), ("up_drop_site", C.c_uint32),
        ("end_y", C.c_void_p), ("end_gamma", C.c_void_p), ("end_mean", C.c_void_p), ("end_rstd", C.c_void_p),
        ("end_lens", C.c_void_p), ("end_T", C.c_int32), ("end_ws", C.c_void_p), ("end_replicas", C.c_int32),
        ("dres_out", C.c_void_p), ("dy_out", C.c_void_p), ("drop_o_p", C.c_float), ("drop_o_site", C.c_uint32),
        ("pair_ws", C.c_void_p), ("pair_ws_bytes", C.c_int64), ("z_tiled", C.c_int32),
    ]


class RowblockDgradArgs(C.Structure):
    """Mirror of ``struct s2t_rowblock_dgrad_args`` (include/s2t_hip.h)."""

    _fields_ = [
        ("dy", C.c_void_p), ("wt", C.c_void_p), ("d", C.c_int32), ("M", C.c_int32), ("K", C.c_int32), ("dxn", C.c_void_p),
        ("ln_x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p),
        ("ln_lens", C.c_void_p), ("ln_T", C.c_int32), ("dres", C.c_void_p), ("ln_ws", C.c_void_p), ("ln_replicas", C.c_int32),
        ("dx", C.c_void_p), ("dx_drop", C.c_void_p), ("up_drop_p", C.c_float), ("up_drop_site", C.c_uint32),
        ("drop_seed", C.c_void_p),
    ]


class RowblockArgs(C.Structure):
    """Mirror of ``struct s2t_rowblock_args`` (include/s2t_hip.h)."""

    _fields_ = [
        ("x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("d", C.c_int32),
        ("ln_lens", C.c_void_p), ("ln_T", C.c_int32),
        ("x_ln", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("act", C.c_int32),
        ("preact", C.c_void_p), ("ldp", C.c_int64), ("out", C.c_void_p), ("ldc", C.c_int64), ("alpha", C.c_float),
        ("row_lens", C.c_void_p), ("row_T", C.c_int32), ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("drop_p", C.c_float), ("drop_site", C.c_uint32), ("drop_seed", C.c_void_p),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_act", C.c_int32),
        ("conv_w", C.c_void_p), ("conv_T", C.c_int32),
        ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("bn_eps", C.c_float),
    ]


_CTYPE = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "float": C.c_float, "uint32_t": C.c_uint32}


RESTYPES = {}


def header_prototypes(path=HEADER_PATH):
    """Parse ``int|int64_t s2t_*(...);`` declarations of the header -> {name: [ctypes argtypes]} (+ RESTYPES)."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(s2t_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        RESTYPES[name] = _CTYPE[m.group(1)]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    if "s2t_gemm_args" in a:
                        argtypes.append(C.POINTER(GemmArgs))
                    elif "s2t_ffn_args" in a:
                        argtypes.append(C.POINTER(FfnArgs))
                    elif "s2t_ffn_bwd_args" in a:
                        argtypes.append(C.POINTER(FfnBwdArgs))
                    elif "s2t_rowblock_dgrad_args" in a:
                        argtypes.append(C.POINTER(RowblockDgradArgs))
                    elif "s2t_rowblock_args" in a:
                        argtypes.append(C.POINTER(RowblockArgs))
                    else:
                        argtypes.append(C.c_void_p)
                else:
                    ty = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPE[ty])
        protos[name] = argtypes
    return protos


def lib():
    """Return the loaded library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "s2t_amd: %s is missing — run `python -m s2t_amd.build` (or __graft_entry__.build()); "
                "there is no CPU fallback for the HIP hot path" % LIB_PATH
            )
        l = C.CDLL(LIB_PATH)
        for name, argtypes in header_prototypes().items():
            fn = getattr(l, name)  # AttributeError here = header declares a symbol the library lacks
            fn.restype = RESTYPES[name]
            fn.argtypes = argtypes
        _lib = l
    return _lib


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
