"""Host-side plumbing for the HIP activation solver: device buffers, streams, workspace.

PyTorch is used for device memory and stream handles only; all arithmetic happens in
libevc_hip.so (hand-written gfx950 kernels) behind the C ABI of include/evc.h.

Orientation (`layout`):
  "bin_major"   - north_star / pymf / nmf_tool: A (M,N), X (M,T), H (N,T), B (Mb,N), Y (Mb,T)
  "frame_major" - the reference scripts (04_align_n_nmf.py): A (N,M), X (T,M), H (T,N),
                  B (N,Mb), Y (T,Mb)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

from . import _lib

_EPS_MODES = {"add": _lib.EPS_ADD, "zero_replace": _lib.EPS_ZERO_REPLACE, "none": _lib.EPS_NONE,
              "clamp": _lib.EPS_CLAMP}
_EPS_DEFAULT = {"add": 1e-9, "zero_replace": float(np.finfo(np.float32).eps), "none": 0.0,
                "clamp": 1e-15}
_ALGOS = {"gram": _lib.ALGO_GRAM, "factored": _lib.ALGO_FACTORED, "literal": _lib.ALGO_LITERAL,
          "auto": _lib.ALGO_AUTO}
_INITS = {"given": _lib.INIT_GIVEN, "sklearn": _lib.INIT_SKLEARN, "const": _lib.INIT_CONST}
_STOPS = {"none": _lib.STOP_NONE, "sklearn": _lib.STOP_SKLEARN, "pymf": _lib.STOP_PYMF}
_LAYOUTS = {"frame_major": _lib.FRAME_MAJOR, "bin_major": _lib.BIN_MAJOR}
_LOSSES = {"frobenius": _lib.LOSS_FROBENIUS, "kullback-leibler": _lib.LOSS_KL, "kl": _lib.LOSS_KL}

# scratch memory handed to the C ABI, one buffer per (device, stream): include/evc.h promises that calls on
# distinct streams are independent, so two streams (or two host threads on two streams) must never share
# scratch.  A buffer belongs to the stream it was allocated on (torch's allocator is stream-ordered), so
# growing it is safe: the old block is only reused by later work of the same stream.
_workspaces = {}
_ws_lock = threading.Lock()


def _torch():
    import torch
    return torch


def require_device(device=None):
    """Resolve the HIP device to run on; raise if there is none (no CPU fallback)."""
    torch = _torch()
    _lib.lib()  # fail early and loudly when the native library is missing
    if not torch.cuda.is_available():
        raise RuntimeError("exemplars_vc_amd needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"exemplars_vc_amd runs on HIP devices only, got {device}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class _Scratch:
    def __init__(self):
        self.lock = threading.Lock()
        self.buf = None


@contextlib.contextmanager
def _workspace(nbytes: int, device):
    """The scratch buffer of (device, current stream), held for the duration of one native call."""
    torch = _torch()
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    with _ws_lock:
        slot = _workspaces.get(key)
        if slot is None:
            slot = _workspaces[key] = _Scratch()
    with slot.lock:
        if slot.buf is None or slot.buf.numel() < nbytes:
            slot.buf = None
            with torch.cuda.device(device):
                slot.buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        yield slot.buf


def release_workspaces():
    with _ws_lock:
        _workspaces.clear()


def _to_dev(x, tdtype, device):
    """2-D array/tensor -> device tensor with unit inner stride; returns (tensor, was_numpy)."""
    torch = _torch()
    was_numpy = not isinstance(x, torch.Tensor)
    if was_numpy:
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if x.dim() != 2:
        raise ValueError(f"expected a 2-D matrix, got shape {tuple(x.shape)}")
    x = x.to(device=device, dtype=tdtype)
    if x.shape[1] > 1 and x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x, was_numpy


def _to_host(t):
    """Device tensor -> numpy.  Large results go through page-locked memory (torch caches the pinned
    blocks): a pageable download of one utterance's activations (22 MB) costs as much as its solve."""
    torch = _torch()
    if t.numel() * t.element_size() < (1 << 20):
        return t.cpu().numpy()
    t = t.contiguous()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _ld(x):
    return int(x.stride(0)) if x.shape[0] > 1 else int(max(x.shape[1], 1))


def _pick_dtype(dtype, *arrs):
    torch = _torch()
    if dtype is None:
        for a in arrs:
            if a is None:
                continue
            dt = a.dtype
            if dt in (np.float32, torch.float32):
                return torch.float32, _lib.F32
            break
        return torch.float64, _lib.F64
    if dtype in ("f32", "float32", np.float32, torch.float32):
        return torch.float32, _lib.F32
    if dtype in ("f64", "float64", np.float64, torch.float64, float):
        return torch.float64, _lib.F64
    raise ValueError(f"unsupported dtype {dtype!r}")


class PreparedDictionary:
    """A dictionary imported once into the layouts the kernels read (evc_dict_prepare): pass it in place of `A` (and
    leave `B` out) to solve_activations / convert.  The reference builds A and B once per run
    (04_align_n_nmf.py:230-246,350-361); results are bitwise those of the unprepared calls."""

    def __init__(self, handle, buf, layout, tdtype, dcode, device, loss, eps):
        self.handle, self._buf = handle, buf           # the device image must outlive the handle's uses
        self.layout, self.tdtype, self.dcode, self.device, self.loss, self.eps = layout, tdtype, dcode, device, loss, eps
        self.M, self.Mb, self.N = int(handle.M), int(handle.Mb), int(handle.N)


def prepare_dictionary(A, B=None, *, layout="bin_major", dtype=None, loss="frobenius", eps=None, device=None):
    """Import A (and the parallel target dictionary B) once; see PreparedDictionary."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    lay = _LAYOUTS[layout]
    tdtype, dcode = _pick_dtype(dtype, A)
    A_d, _ = _to_dev(A, tdtype, device)
    M, N = A_d.shape if lay == _lib.BIN_MAJOR else A_d.shape[::-1]
    Mb, B_d = 0, None
    if B is not None:
        B_d, _ = _to_dev(B, tdtype, device)
        Mb, N2 = B_d.shape if lay == _lib.BIN_MAJOR else B_d.shape[::-1]
        if N2 != N:
            raise ValueError(f"A and B disagree on the number of exemplars: {N} vs {N2}")
    lcode = _LOSSES[loss]
    if eps is None:
        eps = _EPS_DEFAULT["zero_replace"] if lcode == _lib.LOSS_KL else 0.0
    nbytes = int(L.evc_dict_bytes(M, Mb, N, dcode, lcode))
    if nbytes == 0:
        raise ValueError("unsupported dictionary shape / dtype")
    with torch.cuda.device(device):
        buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        off = (-buf.data_ptr()) % 256
        handle = _lib.Dict()
        st = L.evc_dict_prepare(A_d.data_ptr(), _ld(A_d), B_d.data_ptr() if B_d is not None else None,
                                _ld(B_d) if B_d is not None else 0, M, Mb, N, lay, dcode, lcode, float(eps),
                                buf.data_ptr() + off, nbytes, C.byref(handle),
                                C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    _lib.check(st, "evc_dict_prepare")
    # evc.h promises the images to later calls on the SAME stream only; a prepared dictionary is used from side streams
    # and other host threads (compat's three WORLD streams, cached_dictionary hits), so the packing kernels are waited
    # for here, once per dictionary
    torch.cuda.current_stream(device).synchronize()
    return PreparedDictionary(handle, buf, lay, tdtype, dcode, device, lcode, float(eps))


_dict_cache = {}


def cached_dictionary(A, B=None, **kw):
    """prepare_dictionary, remembered per (A, B) object identity: the compat surfaces take plain arrays, and a caller
    converting utterance after utterance against one dictionary passes the same arrays again.  The arrays are held
    weakly; an entry dies with them.  (Content is NOT hashed: mutate a dictionary in place and you must not use this.)"""
    import weakref
    key = (id(A), id(B) if B is not None else None, tuple(getattr(A, "shape", ())), str(getattr(A, "dtype", "")),
           tuple(sorted((k, str(v)) for k, v in kw.items())))
    ent = _dict_cache.get(key)
    if ent is not None and ent[1]() is A and (B is None or ent[2]() is B):
        return ent[0]
    pd = prepare_dictionary(A, B, **kw)
    try:
        ra = weakref.ref(A, lambda _r, k=key: _dict_cache.pop(k, None))
        rb = weakref.ref(B) if B is not None else None
    except TypeError:
        return pd
    if len(_dict_cache) >= 8:
        _dict_cache.pop(next(iter(_dict_cache)))
    _dict_cache[key] = (pd, ra, rb)
    return pd


def workspace_bytes(M, N, T, n_utt=1, dtype="f64", algo="auto", Mb=0):
    _, code = _pick_dtype(dtype)
    return int(_lib.lib().evc_workspace_bytes(M, Mb, N, T, n_utt, code, _ALGOS[algo]))


def solve_activations(A, X, H0=None, **kw):
    """H <- H (.) A^T X (/) guard(A^T A H + l1), `iters` times, on the GPU.

    Returns H in the caller's orientation (numpy in -> numpy out, device tensor in -> device
    tensor out); with info=True also a dict(n_iter=int array per utterance, err=array
    [n_utt, 1+iters//check_every] of residuals, NaN where not evaluated).
    Keywords: see `_solve`.
    """
    return _solve(A, X, H0, None, **kw)


def convert(A, X, B=None, H0=None, *, want_h=True, **kw):
    """Activation solve followed by the synthesis Y = B H in one launch sequence
    (factorize() + convert() of 04_align_n_nmf.py).  Returns (H, Y), or Y alone with
    want_h=False (the activations then never leave the solver's tile layout); with info=True
    the info dict is appended.  `A` may be a PreparedDictionary made with B: then leave B out."""
    if B is None:
        if not (isinstance(A, PreparedDictionary) and A.Mb > 0):
            raise ValueError("convert() needs B (or a prepared dictionary that holds it)")
        B = True
    return _solve(A, X, H0, B, want_h=want_h, **kw)


def _solve(A, X, H0, B, *, layout="bin_major", iters=100, eps_mode="add", eps=None,
           l1=0.0, algo="auto", init=None, init_value=0.0, check_every=0,
           stop_rule="none", tol=0.0, utt_offsets: Optional[Sequence[int]] = None,
           dtype=None, device=None, info=False, out=None, loop_events=None,
           fused=True, fused_c=0, fused_w=0, want_h=True, out_y=None, loss="frobenius", exact_div=False,
           cooperative=True, all_resident=True, pair_tiles=False, _fake_coop_timeout=False, solve_info=None):
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    lay = _LAYOUTS[layout]
    pd = A if isinstance(A, PreparedDictionary) else None
    if pd is not None:
        if dtype is None:
            dtype = pd.tdtype
        if pd.device != device:
            raise ValueError("the prepared dictionary lives on another device")
        A = None
    tdtype, dcode = _pick_dtype(dtype, X, A)
    if pd is not None and (dcode != pd.dcode or _LOSSES[loss] != pd.loss):
        raise ValueError("the prepared dictionary was made for another dtype / loss")
    A_d = None if pd is not None else _to_dev(A, tdtype, device)[0]
    X_d, x_np = _to_dev(X, tdtype, device)
    if lay == _lib.BIN_MAJOR:
        M, N = (pd.M, pd.N) if pd is not None else A_d.shape
        M2, T = X_d.shape
        hshape = (N, T)
    else:
        N, M = (pd.N, pd.M) if pd is not None else A_d.shape
        T, M2 = X_d.shape
        hshape = (T, N)
    if M2 != M:
        raise ValueError(f"A and X disagree on the number of bins: {M} vs {M2}")
    Mb = 0
    B_d = None
    if B is True:                 # convert() with the prepared dictionary's own B
        if pd is None or pd.Mb == 0:
            raise ValueError("convert() without B needs a prepared dictionary that holds B")
        Mb = pd.Mb
    elif B is not None:
        B_d, _ = _to_dev(B, tdtype, device)
        Mb, N2 = B_d.shape if lay == _lib.BIN_MAJOR else B_d.shape[::-1]
        if N2 != N:
            raise ValueError(f"A and B disagree on the number of exemplars: {N} vs {N2}")
    if B is not None:
        yshape = (Mb, T) if lay == _lib.BIN_MAJOR else (T, Mb)
        Y_d = out_y if out_y is not None else torch.empty(yshape, dtype=tdtype, device=device)
    if init is None:
        init = "given" if H0 is not None else "sklearn"
    if init == "given":
        if H0 is None:
            raise ValueError("init='given' needs H0")
        H_d, _ = _to_dev(H0, tdtype, device)
        if tuple(H_d.shape) != hshape:
            raise ValueError(f"H0 has shape {tuple(H_d.shape)}, expected {hshape}")
        if out is not None:
            out.copy_(H_d)
            H_d = out
        elif isinstance(H0, torch.Tensor) and H_d.data_ptr() == H0.data_ptr():
            H_d = H_d.clone()       # never clobber the caller's H0
    elif B is not None and not want_h:
        H_d = None
    else:
        H_d = out if out is not None else torch.empty(hshape, dtype=tdtype, device=device)
    if H_d is not None and (tuple(H_d.shape) != hshape or H_d.dtype != tdtype
                            or (H_d.shape[1] > 1 and H_d.stride(1) != 1)):
        raise ValueError("`out` must be a contiguous device tensor of the activation shape/dtype")

    if utt_offsets is None:
        n_utt, off_arr, off_ptr = 1, None, None
    else:
        off_arr = np.ascontiguousarray(np.asarray(utt_offsets, dtype=np.int32))
        n_utt = len(off_arr) - 1
        if n_utt < 1:
            raise ValueError("utt_offsets needs at least two entries")
        off_ptr = off_arr.ctypes.data_as(C.POINTER(C.c_int))

    opts = _lib.SolveOpts()
    opts.struct_bytes = C.sizeof(_lib.SolveOpts)
    opts.dtype, opts.layout, opts.algo = dcode, lay, _ALGOS[algo]
    opts.iters, opts.eps_mode, opts.init_mode = int(iters), _EPS_MODES[eps_mode], _INITS[init]
    opts.check_every, opts.stop_rule = int(check_every), _STOPS[stop_rule]
    opts.eps = _EPS_DEFAULT[eps_mode] if eps is None else float(eps)
    opts.l1, opts.tol, opts.init_value = float(l1), float(tol), float(init_value)
    opts.loss = _LOSSES[loss]
    # EVC_FLAG_* of include/evc.h: NO_FUSED, EXACT_DIV, NO_EXCHANGE (cooperative=False: no kernel in which
    # workgroups exchange data inside a launch - the call is then fully asynchronous), NO_ALL_RESIDENT, PAIR_TILES
    # (pair_tiles=True: the experimental k_fused_xy instead of k_fused_all); bits
    # 8..15 = 1 or 2 force the general streamed kernel with that many frame tiles per workgroup (tuning; on the wide
    # float32 path: exemplar ranges per frame group), bits 16..19 = wavefronts per workgroup of k_fused_wide (4 / 8)
    opts.reserved = ((0 if fused else _lib.FLAG_NO_FUSED) | (_lib.FLAG_EXACT_DIV if exact_div else 0)
                     | (0 if cooperative else _lib.FLAG_NO_EXCHANGE)
                     | (0 if all_resident else _lib.FLAG_NO_ALL_RESIDENT) | (_lib.FLAG_PAIR_TILES if pair_tiles else 0)
                     | ((int(fused_c) & 0xff) << 8)
                     | ((int(fused_w) & 0xf) << 16))
    if _fake_coop_timeout is not False and _fake_coop_timeout is not None:
        # tests (evc_solve_opts.test_abort_at).  True: the call starts with the abort flag raised; an int k > 0: it is
        # raised in front of the k-th launch of the iteration loop
        opts.test_abort_at = -1 if _fake_coop_timeout is True else int(_fake_coop_timeout)
    sinfo = _lib.SolveInfo()
    sinfo.struct_bytes = C.sizeof(_lib.SolveInfo)
    opts.info = C.addressof(sinfo)
    if pd is not None:
        if loss in ("kl", "kullback-leibler") and opts.eps != pd.eps:
            raise ValueError("the prepared dictionary's KL guard differs from this call's eps")
        opts.dict = C.addressof(pd.handle)
    if loop_events is not None:     # (torch.cuda.Event, torch.cuda.Event), already created
        opts.ev_loop_start = int(loop_events[0].cuda_event)
        opts.ev_loop_stop = int(loop_events[1].cuda_event)

    ws_bytes = int(L.evc_workspace_bytes(M, Mb, N, T, n_utt, dcode, opts.algo))
    n_slots = 1 + (iters // check_every if check_every > 0 else 0)
    n_iter = np.zeros(n_utt, dtype=np.int32) if info else None
    err = np.full((n_utt, n_slots), np.nan) if info else None
    ni_p = n_iter.ctypes.data_as(C.POINTER(C.c_int)) if info else None
    er_p = err.ctypes.data_as(C.POINTER(C.c_double)) if info else None
    h_ptr, h_ld = (H_d.data_ptr(), _ld(H_d)) if H_d is not None else (None, 0)
    a_ptr, a_ld = (A_d.data_ptr(), _ld(A_d)) if A_d is not None else (None, 0)
    with torch.cuda.device(device), _workspace(ws_bytes, device) as ws:
        stream = torch.cuda.current_stream(device).cuda_stream
        if B is None:
            st = L.evc_nmf_solve(
                a_ptr, a_ld, X_d.data_ptr(), _ld(X_d), h_ptr, h_ld,
                M, N, T, off_ptr, n_utt, C.byref(opts), ws.data_ptr(), ws.numel(), ni_p, er_p,
                C.c_void_p(stream))
        else:
            st = L.evc_nmf_convert(
                a_ptr, a_ld, X_d.data_ptr(), _ld(X_d), B_d.data_ptr() if B_d is not None else None,
                _ld(B_d) if B_d is not None else 0,
                h_ptr, h_ld, Y_d.data_ptr(), _ld(Y_d), M, Mb, N, T, off_ptr, n_utt, C.byref(opts),
                ws.data_ptr(), ws.numel(), ni_p, er_p, C.c_void_p(stream))
    _lib.check(st, "evc_nmf_solve" if B is None else "evc_nmf_convert")
    to_np = x_np and out is None
    H_out = None if H_d is None else (_to_host(H_d) if to_np else H_d)
    res = [H_out] if B is None else ([H_out] if want_h else [])
    if B is not None:
        res.append(_to_host(Y_d) if (x_np and out_y is None) else Y_d)
    if info:
        res.append({"n_iter": n_iter, "err": err, "kernel": _lib.KERNEL_NAMES.get(sinfo.kernel, str(sinfo.kernel)),
                    "members": int(sinfo.members), "launches": int(sinfo.launches), "redo": int(sinfo.redo),
                    "exchange": int(sinfo.exchange), "prepared": int(sinfo.prepared)})
    if solve_info is not None:      # caller-supplied dict: filled without the synchronisation info=True implies
        solve_info.update(kernel=_lib.KERNEL_NAMES.get(sinfo.kernel, str(sinfo.kernel)), members=int(sinfo.members),
                          launches=int(sinfo.launches), redo=int(sinfo.redo), exchange=int(sinfo.exchange),
                          prepared=int(sinfo.prepared))
    return res[0] if len(res) == 1 else tuple(res)


def synthesize(B, H, *, layout="bin_major", dtype=None, device=None):
    """Y = B H (bin_major, (Mb,T)) or H B (frame_major: np.matmul(H.T, B) of
    04_align_n_nmf.py:391 with H already frames-as-rows, giving (T,Mb))."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    lay = _LAYOUTS[layout]
    tdtype, dcode = _pick_dtype(dtype, H, B)
    B_d, _ = _to_dev(B, tdtype, device)
    H_d, h_np = _to_dev(H, tdtype, device)
    if lay == _lib.BIN_MAJOR:
        Mb, N = B_d.shape
        N2, T = H_d.shape
        yshape = (Mb, T)
    else:
        N, Mb = B_d.shape
        T, N2 = H_d.shape
        yshape = (T, Mb)
    if N2 != N:
        raise ValueError(f"B and H disagree on the number of exemplars: {N} vs {N2}")
    Y = torch.empty(yshape, dtype=tdtype, device=device)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        st = L.evc_synthesize(B_d.data_ptr(), _ld(B_d), H_d.data_ptr(), _ld(H_d), Y.data_ptr(),
                              _ld(Y), Mb, N, T, lay, dcode, C.c_void_p(stream))
    _lib.check(st, "evc_synthesize")
    return _to_host(Y) if h_np else Y


def frame_residuals(A, X, H, *, layout="bin_major", dtype=None, device=None):
    """Per-frame squared residual sum_m (X - A H)^2 (float64, length T)."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    lay = _LAYOUTS[layout]
    tdtype, dcode = _pick_dtype(dtype, X, A)
    A_d, _ = _to_dev(A, tdtype, device)
    X_d, x_np = _to_dev(X, tdtype, device)
    H_d, _ = _to_dev(H, tdtype, device)
    if lay == _lib.BIN_MAJOR:
        M, N = A_d.shape
        T = X_d.shape[1]
    else:
        N, M = A_d.shape
        T = X_d.shape[0]
    err2 = torch.zeros(max(T, 1), dtype=torch.float64, device=device)
    ws_bytes = int(L.evc_workspace_bytes(M, 0, N, T, 1, dcode, _lib.ALGO_GRAM))
    with torch.cuda.device(device), _workspace(ws_bytes, device) as ws:
        stream = torch.cuda.current_stream(device).cuda_stream
        st = L.evc_residual(A_d.data_ptr(), _ld(A_d), X_d.data_ptr(), _ld(X_d), H_d.data_ptr(),
                            _ld(H_d), M, N, T, lay, dcode, err2.data_ptr(), ws.data_ptr(),
                            ws.numel(), C.c_void_p(stream))
    _lib.check(st, "evc_residual")
    err2 = err2[:T]
    return err2.cpu().numpy() if x_np else err2


def griffin_lim(magnitude_spectrogram, fft_size, hopsamp, iterations, x0, *, device=None, want_rmse=False):
    """Griffin-Lim reconstruction on the GPU (float64): magnitudes (T, fft_size/2+1) with rows as time
    slices and the initial signal x0 (T*hopsamp + fft_size samples) -> reconstructed signal
    [, per-iteration RMSE].  Mirrors zz_audio_utilities.reconstruct_signal_griffin_lim."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    mag, m_np = _to_dev(magnitude_spectrogram, torch.float64, device)
    T, nb = mag.shape
    fft_size, hopsamp, iterations = int(fft_size), int(hopsamp), int(iterations)
    if nb != fft_size // 2 + 1:
        raise ValueError(f"expected {fft_size // 2 + 1} frequency bins for fft_size={fft_size}, got {nb}")
    n = T * hopsamp + fft_size
    was_np = not isinstance(x0, torch.Tensor)
    x = torch.as_tensor(np.asarray(x0, dtype=np.float64) if was_np else x0, dtype=torch.float64).to(device).clone()
    if x.numel() != n:
        raise ValueError(f"x0 must have {n} samples, got {x.numel()}")
    ws_bytes = int(L.evc_griffin_lim_workspace_bytes(T, fft_size, hopsamp, iterations))
    if ws_bytes == 0:
        raise ValueError("unsupported Griffin-Lim configuration (fft_size must be even and >= 2)")
    rmse = np.zeros(max(iterations, 1)) if want_rmse else None
    with torch.cuda.device(device), _workspace(ws_bytes, device) as ws:
        stream = torch.cuda.current_stream(device).cuda_stream
        st = L.evc_griffin_lim(mag.data_ptr(), _ld(mag), T, fft_size, hopsamp, iterations, x.data_ptr(),
                               ws.data_ptr(), ws.numel(),
                               rmse.ctypes.data_as(C.POINTER(C.c_double)) if want_rmse else None,
                               C.c_void_p(stream))
    _lib.check(st, "evc_griffin_lim")
    out = x.cpu().numpy() if (m_np and was_np) else x
    return (out, rmse[:iterations]) if want_rmse else out


def griffin_lim_batch(magnitude_spectrograms, fft_size, hopsamp, iterations, x0s, *, device=None, want_rmse=False):
    """Griffin-Lim for several utterances in ONE call (evc_griffin_lim_batch): a list of (T_u, fft_size/2+1)
    magnitude arrays and the list of their initial signals (T_u*hopsamp + fft_size samples each) -> list of
    reconstructed signals [, (n_utt, iterations) RMSE].  One utterance leaves two thirds of the GPU idle; a batch
    does not.  Results equal per-utterance griffin_lim() calls up to the summation order of the contractions."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    fft_size, hopsamp, iterations = int(fft_size), int(hopsamp), int(iterations)
    n_utt = len(magnitude_spectrograms)
    if n_utt == 0 or len(x0s) != n_utt:
        raise ValueError("need as many initial signals as magnitude spectrograms (at least one)")
    nb = fft_size // 2 + 1
    mags = [_to_dev(m, torch.float64, device) for m in magnitude_spectrograms]
    all_np = all(np_ for _, np_ in mags) and not any(isinstance(x, torch.Tensor) for x in x0s)
    off = [0]
    for m, _ in mags:
        if m.dim() != 2 or m.shape[1] != nb:
            raise ValueError(f"expected (T, {nb}) magnitudes for fft_size={fft_size}, got {tuple(m.shape)}")
        off.append(off[-1] + int(m.shape[0]))
    mag = torch.cat([m for m, _ in mags], dim=0).contiguous()
    lens = [(off[u + 1] - off[u]) * hopsamp + fft_size for u in range(n_utt)]
    xs = []
    for u, x0 in enumerate(x0s):
        x = torch.as_tensor(x0 if isinstance(x0, torch.Tensor) else np.asarray(x0, dtype=np.float64),
                            dtype=torch.float64).to(device).reshape(-1)
        if x.numel() != lens[u]:
            raise ValueError(f"x0 of utterance {u} must have {lens[u]} samples, got {x.numel()}")
        xs.append(x)
    x = torch.cat(xs)
    offs = (C.c_int * (n_utt + 1))(*off)
    ws_bytes = int(L.evc_griffin_lim_batch_workspace_bytes(offs, n_utt, fft_size, hopsamp, iterations))
    if ws_bytes == 0:
        raise ValueError("unsupported Griffin-Lim configuration (fft_size must be even and >= 2, at least one frame)")
    rmse = np.zeros((n_utt, max(iterations, 1))) if want_rmse else None
    with torch.cuda.device(device), _workspace(ws_bytes, device) as ws:
        stream = torch.cuda.current_stream(device).cuda_stream
        st = L.evc_griffin_lim_batch(mag.data_ptr(), _ld(mag), offs, n_utt, fft_size, hopsamp, iterations,
                                     x.data_ptr(), ws.data_ptr(), ws.numel(),
                                     rmse.ctypes.data_as(C.POINTER(C.c_double)) if want_rmse else None,
                                     C.c_void_p(stream))
    _lib.check(st, "evc_griffin_lim_batch")
    outs = list(torch.split(x, lens))
    if all_np:
        outs = [o.cpu().numpy() for o in outs]
    return (outs, rmse[:, :iterations]) if want_rmse else outs


def stft(y, n_fft=400, hop_length=80, *, center=True, device=None):
    """STFT front end on the GPU (float64): samples y -> (re, im), each (n_frames, n_fft/2+1) with rows
    as time slices - the transposed layout the scripts store (`lbr.core.stft(...).T`,
    04_align_n_nmf.py:422-427).  Periodic Hann window, reflect-padded centred frames (librosa's
    defaults)."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    was_np = not isinstance(y, torch.Tensor)
    x = torch.as_tensor(np.asarray(y, dtype=np.float64) if was_np else y, dtype=torch.float64).to(device).contiguous()
    if x.dim() != 1:
        raise ValueError("y must be one-dimensional")
    n, n_fft, hop_length = int(x.numel()), int(n_fft), int(hop_length)
    if n_fft < 2 or n_fft % 2 or hop_length < 1:
        raise ValueError("n_fft must be even and >= 2, hop_length >= 1")
    nb = n_fft // 2 + 1
    T = int(L.evc_stft_frames(n, n_fft, hop_length, int(center))) if n else 0
    re = torch.empty(T, nb, dtype=torch.float64, device=device)
    im = torch.empty(T, nb, dtype=torch.float64, device=device)
    if T:
        with torch.cuda.device(device), \
                _workspace(int(L.evc_stft_workspace_bytes(n, n_fft, hop_length, int(center))), device) as ws:
            stream = torch.cuda.current_stream(device).cuda_stream
            st = L.evc_stft(x.data_ptr(), n, n_fft, hop_length, int(center), re.data_ptr(), nb, im.data_ptr(), nb,
                            ws.data_ptr(), ws.numel(), C.c_void_p(stream))
        _lib.check(st, "evc_stft")
    return (re.cpu().numpy(), im.cpu().numpy()) if was_np else (re, im)


def dtw_dictionary(dtw_a, dtw_b, src_feats, tar_feats, *, op="copy", real_part=False, dtype=None, device=None):
    """DTW alignment of parallel utterance pairs AND the gather of the aligned frames, on the GPU with no round trip of
    frames through the host (01_make_dict_parallel.py:215-249, 04_align_n_nmf.py:100-169,230-246,320-324): the paths stay
    on the device, an exclusive scan of their lengths places every pair in the dictionary, one kernel copies the rows.

      dtw_a[p], dtw_b[p]        : (frames, features) float64 - what the alignment is computed on (the script: MFCCs)
      src_feats[p], tar_feats[p]: (frames_a, cols) / (frames_b, cols_b) - the frames the dictionary is made of; complex
                                  arrays with real_part=True contribute their real parts (the script's `real`)
      op                        : "copy" | "abs" (the STFT flow stacks np.abs of the real parts, :320-324)
    Returns (A, B, row_start): device tensors N x cols and N x cols_b (frames / exemplars as rows, the orientation
    `prepare_dictionary(layout="frame_major")` and the solver take) and the first row of every pair (numpy, n_pairs + 1).
    Only N (one int) is read back - it sizes the dictionary."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    n = len(dtw_a)
    if n == 0 or not (len(dtw_b) == len(src_feats) == len(tar_feats) == n):
        raise ValueError("need the same, non-zero number of utterances in every list")
    dtw_a = [np.ascontiguousarray(np.asarray(f, dtype=np.float64)) for f in dtw_a]
    dtw_b = [np.ascontiguousarray(np.asarray(f, dtype=np.float64)) for f in dtw_b]
    D = dtw_a[0].shape[1]
    if any(f.ndim != 2 or f.shape[1] != D for f in dtw_a + dtw_b):
        raise ValueError("every utterance must be (frames, features) with the same number of features")
    for fa, fb, sa, sb in zip(dtw_a, dtw_b, src_feats, tar_feats):
        if len(sa) != len(fa) or len(sb) != len(fb):
            raise ValueError("alignment features and dictionary frames must have the same number of frames per utterance")
    aoff = np.concatenate([[0], np.cumsum([len(f) for f in dtw_a])]).astype(np.int32)
    boff = np.concatenate([[0], np.cumsum([len(f) for f in dtw_b])]).astype(np.int32)

    def stack(feats):
        m = np.concatenate([np.asarray(f) for f in feats], axis=0)
        if np.iscomplexobj(m):
            if not real_part:
                raise ValueError("complex frames need real_part=True")
            base = np.float32 if m.dtype == np.complex64 else np.float64
            v = np.ascontiguousarray(m).view(base)                 # interleaved re / im: the kernel strides by 2
            return v, 2, m.shape[1]
        base = np.float32 if (m.dtype == np.float32 and dtype != "f64") else np.float64
        return np.ascontiguousarray(m.astype(base, copy=False)), 1, m.shape[1]

    sa, stride_a, cols_a = stack(src_feats)
    sb, stride_b, cols_b = stack(tar_feats)
    if sa.dtype != sb.dtype:
        sa, sb = sa.astype(np.float64), sb.astype(np.float64)
    tdt = torch.float64 if sa.dtype == np.float64 else torch.float32
    dcode = _lib.F64 if sa.dtype == np.float64 else _lib.F32
    opc = {"copy": 0, "abs": 1}[op]
    ap, bp = aoff.ctypes.data_as(C.POINTER(C.c_int)), boff.ctypes.data_as(C.POINTER(C.c_int))
    ws_bytes = int(L.evc_dtw_workspace_bytes(ap, bp, n))
    if ws_bytes == 0:
        raise ValueError("utterance too long for the DTW kernel's wavefront buffers")
    cap = int(aoff[-1] + boff[-1])
    with torch.cuda.device(device):
        FA = torch.from_numpy(np.concatenate(dtw_a, axis=0)).to(device)
        FB = torch.from_numpy(np.concatenate(dtw_b, axis=0)).to(device)
        SA, SB = torch.from_numpy(sa).to(device), torch.from_numpy(sb).to(device)
        pa = torch.empty(max(cap, 1), dtype=torch.int32, device=device)
        pb = torch.empty(max(cap, 1), dtype=torch.int32, device=device)
        plen = torch.empty(n, dtype=torch.int32, device=device)
        rows = torch.empty(n + 1, dtype=torch.int32, device=device)
        d_aoff = torch.from_numpy(aoff[:-1].copy()).to(device)
        d_boff = torch.from_numpy(boff[:-1].copy()).to(device)
        d_poff = torch.from_numpy((aoff[:-1] + boff[:-1]).astype(np.int32)).to(device)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        with _workspace(ws_bytes, device) as ws:
            st = L.evc_dtw_align(FA.data_ptr(), D, ap, FB.data_ptr(), D, bp, D, n, pa.data_ptr(), pb.data_ptr(),
                                 plen.data_ptr(), None, ws.data_ptr(), ws.numel(), stream)
        _lib.check(st, "evc_dtw_align")
        n_rows = C.c_int(0)
        _lib.check(L.evc_dtw_path_rows(plen.data_ptr(), n, rows.data_ptr(), C.byref(n_rows), stream), "evc_dtw_path_rows")
        N = int(n_rows.value)
        A = torch.empty((N, cols_a), dtype=tdt, device=device)
        B = torch.empty((N, cols_b), dtype=tdt, device=device)
        _lib.check(L.evc_dtw_gather_rows(SA.data_ptr(), SA.shape[1], stride_a, pa.data_ptr(), plen.data_ptr(),
                                         d_aoff.data_ptr(), d_poff.data_ptr(), rows.data_ptr(), n, cols_a, opc,
                                         A.data_ptr(), cols_a, dcode, stream), "evc_dtw_gather_rows")
        _lib.check(L.evc_dtw_gather_rows(SB.data_ptr(), SB.shape[1], stride_b, pb.data_ptr(), plen.data_ptr(),
                                         d_boff.data_ptr(), d_poff.data_ptr(), rows.data_ptr(), n, cols_b, opc,
                                         B.data_ptr(), cols_b, dcode, stream), "evc_dtw_gather_rows")
    return A, B, rows.cpu().numpy()


def dtw_align(feats_a, feats_b, *, device=None, want_cost=False):
    """DTW paths of parallel utterance pairs on the GPU.  feats_a[p], feats_b[p]: (frames, features)
    float64 arrays of pair p.  Returns a list of (path_a, path_b) int arrays [, accumulated costs]:
    what `dtw.dtw(x, y, dist=lambda u, v: sum(np.square(u - v)))[3]` returns per pair."""
    torch = _torch()
    device = require_device(device)
    L = _lib.lib()
    feats_a = [np.ascontiguousarray(np.asarray(f, dtype=np.float64)) for f in feats_a]
    feats_b = [np.ascontiguousarray(np.asarray(f, dtype=np.float64)) for f in feats_b]
    n = len(feats_a)
    if n == 0 or len(feats_b) != n:
        raise ValueError("need the same, non-zero number of utterances on both sides")
    D = feats_a[0].shape[1]
    if any(f.ndim != 2 or f.shape[1] != D for f in feats_a + feats_b):
        raise ValueError("every utterance must be (frames, features) with the same number of features")
    aoff = np.concatenate([[0], np.cumsum([len(f) for f in feats_a])]).astype(np.int32)
    boff = np.concatenate([[0], np.cumsum([len(f) for f in feats_b])]).astype(np.int32)
    A = torch.from_numpy(np.concatenate(feats_a, axis=0)).to(device)
    B = torch.from_numpy(np.concatenate(feats_b, axis=0)).to(device)
    ap, bp = aoff.ctypes.data_as(C.POINTER(C.c_int)), boff.ctypes.data_as(C.POINTER(C.c_int))
    ws_bytes = int(L.evc_dtw_workspace_bytes(ap, bp, n))
    if ws_bytes == 0:
        raise ValueError("utterance too long for the DTW kernel's wavefront buffers")
    cap = int(aoff[-1] + boff[-1])
    pa = torch.empty(max(cap, 1), dtype=torch.int32, device=device)
    pb = torch.empty(max(cap, 1), dtype=torch.int32, device=device)
    plen = torch.empty(n, dtype=torch.int32, device=device)
    tot = torch.empty(n, dtype=torch.float64, device=device)
    with torch.cuda.device(device), _workspace(ws_bytes, device) as ws:
        stream = torch.cuda.current_stream(device).cuda_stream
        st = L.evc_dtw_align(A.data_ptr(), D, ap, B.data_ptr(), D, bp, D, n, pa.data_ptr(), pb.data_ptr(),
                             plen.data_ptr(), tot.data_ptr(), ws.data_ptr(), ws.numel(), C.c_void_p(stream))
    _lib.check(st, "evc_dtw_align")
    pa, pb, plen = pa.cpu().numpy(), pb.cpu().numpy(), plen.cpu().numpy()
    paths = []
    for p in range(n):
        o = int(aoff[p] + boff[p])
        paths.append((pa[o:o + plen[p]].astype(np.int64), pb[o:o + plen[p]].astype(np.int64)))
    return (paths, tot.cpu().numpy()) if want_cost else paths
