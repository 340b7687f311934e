"""ctypes binding of libevc_hip.so (the C ABI declared in include/evc.h).

The product path has no CPU fallback: if the library is missing or cannot be loaded,
`lib()` raises and every solver entry point fails with it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVC_LIB overrides the library path (A/B builds during kernel tuning)
LIB_PATH = os.environ.get("EVC_LIB") or os.path.join(_HERE, "libevc_hip.so")

F64, F32 = 0, 1
FRAME_MAJOR, BIN_MAJOR = 0, 1
EPS_ADD, EPS_ZERO_REPLACE, EPS_NONE, EPS_CLAMP = 0, 1, 2, 3
ALGO_GRAM, ALGO_FACTORED, ALGO_LITERAL, ALGO_AUTO = 0, 1, 2, 3
INIT_GIVEN, INIT_SKLEARN, INIT_CONST = 0, 1, 2
STOP_NONE, STOP_SKLEARN, STOP_PYMF = 0, 1, 2
LOSS_FROBENIUS, LOSS_KL = 0, 1
FLAG_NO_FUSED, FLAG_EXACT_DIV, FLAG_NO_EXCHANGE, FLAG_NO_ALL_RESIDENT, FLAG_PAIR_TILES = 1, 2, 4, 16, 32

# every symbol include/evc.h declares; tests check that the library exports all of them
SYMBOLS = ("evc_version", "evc_strerror", "evc_device_count", "evc_workspace_bytes",
           "evc_nmf_solve", "evc_nmf_convert", "evc_synthesize", "evc_residual",
           "evc_griffin_lim_workspace_bytes", "evc_griffin_lim", "evc_griffin_lim_batch_workspace_bytes",
           "evc_griffin_lim_batch", "evc_dtw_workspace_bytes", "evc_dtw_align",
           "evc_stft_frames", "evc_stft_workspace_bytes", "evc_stft", "evc_dict_bytes", "evc_dict_prepare",
           "evc_dtw_path_rows", "evc_dtw_gather_rows")


class SolveOpts(C.Structure):
    """Mirror of `evc_solve_opts` (include/evc.h)."""
    _fields_ = [
        ("struct_bytes", C.c_int), ("dtype", C.c_int), ("layout", C.c_int), ("algo", C.c_int),
        ("iters", C.c_int), ("eps_mode", C.c_int), ("init_mode", C.c_int),
        ("check_every", C.c_int), ("stop_rule", C.c_int), ("reserved", C.c_int),
        ("loss", C.c_int), ("test_abort_at", C.c_int),
        ("eps", C.c_double), ("l1", C.c_double), ("tol", C.c_double), ("init_value", C.c_double),
        ("ev_loop_start", C.c_void_p), ("ev_loop_stop", C.c_void_p), ("info", C.c_void_p), ("dict", C.c_void_p),
    ]


class Dict(C.Structure):
    """Mirror of `evc_dict` (include/evc.h): a dictionary imported once (evc_dict_prepare)."""
    _fields_ = [
        ("struct_bytes", C.c_int), ("magic", C.c_int), ("M", C.c_int), ("Mb", C.c_int), ("N", C.c_int),
        ("dtype", C.c_int), ("loss", C.c_int), ("reserved", C.c_int),
        ("eps", C.c_double), ("mem", C.c_void_p), ("bytes", C.c_size_t),
    ]


class SolveInfo(C.Structure):
    """Mirror of `evc_solve_info` (include/evc.h): what the library actually ran."""
    _fields_ = [
        ("struct_bytes", C.c_int), ("kernel", C.c_int), ("members", C.c_int), ("launches", C.c_int),
        ("redo", C.c_int), ("exchange", C.c_int), ("prepared", C.c_int), ("reserved", C.c_int),
    ]


KERNEL_NAMES = {0: "none", 1: "k_gemm_nt", 2: "k_gemm2", 3: "k_fused_mu", 4: "k_fused_res", 5: "k_fused_all",
                6: "k_fused_wide", 7: "k_fused_wide64", 8: "k_fused_xy"}


class EvcError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"{what}: {strerror(status)} (status {status})")
        self.status = status


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m exemplars_vc_amd.csrc.build` "
            "(hipcc, gfx950).  exemplars_vc_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; import it FIRST so that
    # libevc_hip.so binds to the HIP runtime torch already initialised (one runtime per process:
    # shared device context, interoperable streams and events).  Loaded the other way round, two
    # HSA runtimes end up in the process and the second one sees no device.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.evc_version.restype = C.c_int
    L.evc_strerror.restype = C.c_char_p
    L.evc_strerror.argtypes = [C.c_int]
    L.evc_device_count.restype = C.c_int
    L.evc_workspace_bytes.restype = C.c_size_t
    L.evc_workspace_bytes.argtypes = [C.c_int] * 7
    L.evc_nmf_solve.restype = C.c_int
    L.evc_nmf_solve.argtypes = [
        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,      # A, X, H
        C.c_int, C.c_int, C.c_int,                                          # M, N, T
        C.POINTER(C.c_int), C.c_int,                                        # utt_offsets, n_utt
        C.POINTER(SolveOpts),
        C.c_void_p, C.c_size_t,                                             # workspace
        C.POINTER(C.c_int), C.POINTER(C.c_double),                          # n_iter_out, err_out
        C.c_void_p,                                                         # stream
    ]
    L.evc_nmf_convert.restype = C.c_int
    L.evc_nmf_convert.argtypes = [
        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,      # A, X, B
        C.c_void_p, C.c_int, C.c_void_p, C.c_int,                           # H, Y
        C.c_int, C.c_int, C.c_int, C.c_int,                                 # M, Mb, N, T
        C.POINTER(C.c_int), C.c_int, C.POINTER(SolveOpts),
        C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p,
    ]
    L.evc_dict_bytes.restype = C.c_size_t
    L.evc_dict_bytes.argtypes = [C.c_int] * 5
    L.evc_dict_prepare.restype = C.c_int
    L.evc_dict_prepare.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.POINTER(Dict), C.c_void_p]
    L.evc_synthesize.restype = C.c_int
    L.evc_synthesize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.evc_residual.restype = C.c_int
    L.evc_residual.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.evc_griffin_lim_workspace_bytes.restype = C.c_size_t
    L.evc_griffin_lim_workspace_bytes.argtypes = [C.c_int] * 4
    L.evc_griffin_lim.restype = C.c_int
    L.evc_griffin_lim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_void_p]
    L.evc_griffin_lim_batch_workspace_bytes.restype = C.c_size_t
    L.evc_griffin_lim_batch_workspace_bytes.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
    L.evc_griffin_lim_batch.restype = C.c_int
    L.evc_griffin_lim_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_void_p]
    L.evc_stft_frames.restype = C.c_int
    L.evc_stft_frames.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int]
    L.evc_stft_workspace_bytes.restype = C.c_size_t
    L.evc_stft_workspace_bytes.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int]
    L.evc_stft.restype = C.c_int
    L.evc_stft.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                           C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.evc_dtw_workspace_bytes.restype = C.c_size_t
    L.evc_dtw_workspace_bytes.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.evc_dtw_align.restype = C.c_int
    L.evc_dtw_align.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p]
    L.evc_dtw_path_rows.restype = C.c_int
    L.evc_dtw_path_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    L.evc_dtw_gather_rows.restype = C.c_int
    L.evc_dtw_gather_rows.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    _lib = L
    return L


def strerror(status: int) -> str:
    return lib().evc_strerror(int(status)).decode()


def check(status: int, what: str):
    if status != 0:
        raise EvcError(status, what)
