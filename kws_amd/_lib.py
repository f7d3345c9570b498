"""ctypes binding of libfastgrnn_hip.so (C ABI: include/fastgrnn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C kws_amd/csrc``).
There is NO fallback: if the shared object is missing or does not export the ABI
this module raises, and every operator in ``kws_amd`` fails with it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FASTGRNN_HIP_LIB: A/B a differently built library (tools/) without touching the in-tree one
LIB_PATH = os.environ.get("FASTGRNN_HIP_LIB") or os.path.join(_HERE, "csrc", "libfastgrnn_hip.so")

ABI_VERSION = 1

F32, F64, BF16_IO = 0, 1, 2     # fastgrnn_dtype; BF16_IO: bf16 sequences, fp32 everything else
FLAG_FORCE_GENERIC = 1
FLAG_FORCE_F32_MFMA = 2
FLAG_SAVE_PREACT = 4
FLAG_FWD_4WAVE = 8            # A/B: older 4-wave forward shape
FLAG_FWD_BF16X3 = 64          # A/B: forward state product on three bf16 planes
FLAG_X_BFT = 128              # x / d_x are the trainer's [B,F,T]
FLAG_GRAD_LAST = 256          # backward_unroll: grad_h is [B,H], the gradient of the last state alone (model.py:227)
FLAG_HS_LAST = 512            # forward_unroll (inference, no saved tensors): hs is [B,H] = h_T
FLAG_BATCH_MAJOR = 16         # sequences are [B,T,.] (batch_first) instead of [T,B,.]

# include/fastgrnn_hip.h: fastgrnn_nonlinearity.  0..2 are the reference's table
# (rnn.py:478,751); 3..5 the CPU cell's quantised family (rnn.py:53-60).
NONLINEARITY = {"sigmoid": 0, "relu": 1, "tanh": 2, "quantTanh": 3, "quantSigm": 4, "quantSigm4": 5}

EXPORTS = (
    "fastgrnn_hip_abi_version", "fastgrnn_hip_status_string", "fastgrnn_hip_kernel_path",
    "fastgrnn_hip_forward_workspace_bytes", "fastgrnn_hip_backward_workspace_bytes",
    "fastgrnn_hip_forward_unroll", "fastgrnn_hip_backward_unroll",
    "fastgrnn_hip_forward", "fastgrnn_hip_backward",
    "fastgrnn_hip_head_workspace_bytes", "fastgrnn_hip_head_xent", "fastgrnn_hip_debug_poison_cu_state",
    "fastgrnn_hip_frame_gemm",
)


class Desc(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("F", C.c_int32), ("H", C.c_int32),
                ("w_rank", C.c_int32), ("u_rank", C.c_int32),
                ("gate_nl", C.c_int32), ("update_nl", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("w", "u", "w1", "w2", "u1", "u2", "bias_gate", "bias_update", "zeta", "nu")]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("d_x", "d_bias_gate", "d_bias_update", "d_zeta", "d_nu", "d_h0",
                 "d_w", "d_u", "d_w1", "d_w2", "d_u1", "d_u2")]


class FastGRNNLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes handle; raise loudly if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FastGRNNLibraryError(
            "libfastgrnn_hip.so not found at %s -- build it first: "
            "python -c 'import __graft_entry__ as g; g.build()'  (or make -C kws_amd/csrc). "
            "kws_amd has no CPU or eager fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise FastGRNNLibraryError("%s does not export %s" % (LIB_PATH, name))
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    DP, PP, GP = C.POINTER(Desc), C.POINTER(Params), C.POINTER(Grads)
    lib.fastgrnn_hip_abi_version.restype = i32
    lib.fastgrnn_hip_status_string.restype = C.c_char_p
    lib.fastgrnn_hip_status_string.argtypes = [i32]
    lib.fastgrnn_hip_kernel_path.restype = i32
    lib.fastgrnn_hip_kernel_path.argtypes = [DP, i32]
    for f in (lib.fastgrnn_hip_forward_workspace_bytes, lib.fastgrnn_hip_backward_workspace_bytes):
        f.restype = sz
        f.argtypes = [DP]
    lib.fastgrnn_hip_forward_unroll.restype = i32
    lib.fastgrnn_hip_forward_unroll.argtypes = [DP, PP, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.fastgrnn_hip_backward_unroll.restype = i32
    lib.fastgrnn_hip_backward_unroll.argtypes = [DP, PP, vp, vp, vp, vp, vp, vp, GP, vp, sz, vp]
    lib.fastgrnn_hip_forward.restype = i32
    lib.fastgrnn_hip_forward.argtypes = [DP, PP, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.fastgrnn_hip_backward.restype = i32
    lib.fastgrnn_hip_backward.argtypes = [DP, PP, vp, vp, vp, vp, vp, GP, vp, sz, vp]
    lib.fastgrnn_hip_debug_poison_cu_state.restype = i32
    lib.fastgrnn_hip_debug_poison_cu_state.argtypes = [C.c_uint32, vp]
    lib.fastgrnn_hip_head_workspace_bytes.restype = sz
    lib.fastgrnn_hip_head_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.fastgrnn_hip_head_xent.restype = i32
    lib.fastgrnn_hip_head_xent.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.fastgrnn_hip_frame_gemm.restype = i32
    lib.fastgrnn_hip_frame_gemm.argtypes = [sz, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp]
    if lib.fastgrnn_hip_abi_version() != ABI_VERSION:
        raise FastGRNNLibraryError("ABI version mismatch: library %d, binding %d"
                                   % (lib.fastgrnn_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def status_string(code):
    return load().fastgrnn_hip_status_string(int(code)).decode()


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (fastgrnn_status %d)" % (what, status_string(code), code))
