"""ctypes binding of libtinynerf_hip.so (C ABI: include/tinynerf_hip.h).

The product path has no CPU fallback: if the library is missing the first call raises
``RuntimeError``.  torch is only used for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TN_LIB_PATH") or os.path.join(_HERE, "libtinynerf_hip.so")      # (TN_LIB_PATH: A/B of two builds on one box)

TN_MLP_MAX_LAYERS = 12
TN_KPLANES_MAX_SCALES = 4

MARCH_AABB, MARCH_UNBOUNDED = 0, 1
CONTRACT_AABB, CONTRACT_MIP360_INF, CONTRACT_MIP360_L2 = 0, 1, 2
ACT_NONE, ACT_EXP_M1, ACT_SIGMOID, ACT_EXP = 0, 1, 2, 3
ENC_NONE, ENC_POSENC, ENC_DIR_CAT, ENC_AUX_CAT = 0, 1, 2, 3


class SamplerDesc(C.Structure):
    _fields_ = [
        ("marcher", C.c_int32), ("contraction", C.c_int32), ("n_samples", C.c_int32),
        ("grid_d", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32),
        ("aabb", C.c_float * 6), ("near", C.c_float), ("far", C.c_float),
        ("step_size", C.c_float), ("threshold", C.c_float),
        ("t_table", C.c_void_p), ("delta_table", C.c_void_p), ("grid", C.c_void_p), ("jitter", C.c_void_p),
        ("seed", C.c_uint64), ("use_rng", C.c_int32), ("reserved", C.c_int32),
        ("coarse", C.c_void_p),
    ]


class MlpDesc(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("in_dim", C.c_int32), ("dims", C.c_int32 * (TN_MLP_MAX_LAYERS + 1)),
        ("encoding", C.c_int32), ("n_freqs", C.c_int32), ("out_activation", C.c_int32), ("flags", C.c_int32),
        ("freqs", C.c_void_p),
        ("weights", C.c_void_p * TN_MLP_MAX_LAYERS), ("biases", C.c_void_p * TN_MLP_MAX_LAYERS),
        ("aux_index", C.c_void_p), ("aux_stride", C.c_int32), ("reserved", C.c_int32),
        ("row_gate", C.c_void_p),
        ("x_rows", C.c_void_p), ("grad_x_rows", C.c_void_p), ("x_rows_tile_stride", C.c_int64), ("grad_x_rows_tile_stride", C.c_int64),
        ("grad_x_mask_rows", C.c_void_p), ("grad_x_mask_tile_stride", C.c_int64),
    ]


class MergeHead(C.Structure):
    _fields_ = [
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("rows", C.c_int32), ("ld", C.c_int32), ("col0", C.c_int32), ("reserved", C.c_int32),
        ("out_weight", C.c_void_p), ("out_bias", C.c_void_p),
        ("grad_merged_weight", C.c_void_p), ("grad_merged_bias", C.c_void_p),
    ]


class KPlanesDesc(C.Structure):
    _fields_ = [
        ("n_scales", C.c_int32), ("channels", C.c_int32),
        ("height", C.c_int32 * TN_KPLANES_MAX_SCALES), ("width", C.c_int32 * TN_KPLANES_MAX_SCALES),
        ("planes", (C.c_void_p * 3) * TN_KPLANES_MAX_SCALES),
    ]


TN_COBAFA_MAX_LEVELS = 8
TN_MULTI_MAX = 32
MLP_ACCUM_GRAD_X = 1
MLP_STASHED = 2
MLP_CHAIN_ONLY = 4
MLP_WGRAD_ONLY = 8
MLP_GRAD_Y_ROWS = 16
MLP_BF16X3 = 32
MLP_F16X2 = 64
MLP_ROWS_ONLY = 128
MLP_X_FROM_ROWS = 256
MLP_LEAN = 512
MLP_SKIP_LAST = 1024
MLP_LAYERWISE = 2048
ABI_VERSION = 6            # include/tinynerf_hip.h TN_ABI_VERSION: a stale library (TN_LIB_PATH, a forgotten rebuild) fails at load, not in a kernel


class PlaneRegItem(C.Structure):
    _fields_ = [("plane", C.c_void_p), ("grad", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("cy", C.c_float), ("cx", C.c_float), ("cl1", C.c_float)]


class AdamItem(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64)]



class AdamRegItem(C.Structure):
    _fields_ = [("param", C.c_void_p), ("param_out", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("sum_slot", C.c_int32),
                ("cy", C.c_float), ("cx", C.c_float), ("cl1", C.c_float), ("row0", C.c_int32), ("row1", C.c_int32), ("reserved", C.c_int32)]


class CobafaDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32), ("coef_res", C.c_int32 * 3), ("res", (C.c_int32 * 3) * TN_COBAFA_MAX_LEVELS),
        ("channels", C.c_int32 * TN_COBAFA_MAX_LEVELS), ("freqs", C.c_float * TN_COBAFA_MAX_LEVELS),
        ("coef", C.c_void_p), ("basis", C.c_void_p * TN_COBAFA_MAX_LEVELS),
    ]


_lib: Optional[C.CDLL] = None
_TRACE = bool(os.environ.get("TN_TRACE"))


def lib() -> C.CDLL:
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP kernels were not built. Run `python -m tinynerf_amd.build` "
                "(needs hipcc). tinynerf_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.tn_last_error_string.restype = C.c_char_p
        got = int(_lib.tn_abi_version())
        if got != ABI_VERSION:
            _lib = None
            raise RuntimeError(f"{LIB_PATH} has ABI version {got}, tinynerf_amd expects {ABI_VERSION}: rebuild it (`python -m tinynerf_amd.build`)")
    return _lib


def check(rc: int, fn: str) -> None:
    if rc != 0:
        msg = lib().tn_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{fn} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(None) if t is None else C.c_void_p(t.data_ptr())


def stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors: Optional[torch.Tensor]) -> torch.device:
    """Mirror of the reference's CHECK_CUDA / CHECK_CONTIGUOUS (src/cuda.cu:62-64)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("tinynerf_amd: tensor must be a CUDA (HIP) tensor -- there is no CPU path")
        if not t.is_contiguous():
            raise RuntimeError("tinynerf_amd: tensor must be contiguous")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("tinynerf_amd: tensors are on different devices")
    assert dev is not None
    return dev


def call_plain(name: str, *args) -> None:
    """Invoke a host-only entry point (no stream argument); raises on a non-zero return code."""
    check(getattr(lib(), name)(*args), name)


def call(name: str, device: torch.device, *args) -> None:
    """Invoke entry point `name` on `device`'s current stream (appended as the last argument)."""
    fn = getattr(lib(), name)
    with torch.cuda.device(device):
        rc = fn(*args, stream(device))
    check(rc, name)
    if _TRACE:                      # TN_TRACE=1: synchronise after every launch and name it (fault bisection)
        import sys
        print("tn:", name, file=sys.stderr, flush=True)
        torch.cuda.synchronize(device)
