"""S1/S4: the live script's activation solve and synthesis, on the GPU.

Mirrors /root/reference/04_align_n_nmf.py:
  _factorize(X, W, beta_loss, tol)   :194-215   (sklearn non_negative_factorization,
                                                 update_H=False, solver='mu', max_iter=150)
  convert(...) STFT branch           :385-393   (np.matmul(H.T, B))
Argument meaning, return orientation, errors and warnings follow the reference call
(scikit-learn's `_check_init`, dtype check and ConvergenceWarning, _nmf.py:68-82,1221-1226,
1727-1732).
"""
from __future__ import annotations

import os
import warnings

import numpy as np

from ..solver import convert as _solve_and_synthesize, solve_activations, synthesize

try:  # same warning class the reference would raise, when scikit-learn is present
    from sklearn.exceptions import ConvergenceWarning
except Exception:  # pragma: no cover
    class ConvergenceWarning(UserWarning):
        pass

MAX_ITER = 150          # 04_align_n_nmf.py:213
CHECK_EVERY = 10        # sklearn _nmf.py:871


def _check_dictionary(W, n_features):
    W = np.asarray(W)
    if W.ndim != 2:
        raise ValueError(f"Expected 2D array, got {W.ndim}D array instead")
    if W.shape[1] != n_features:
        raise ValueError("Array with wrong second dimension passed to NMF (input H). "
                         f"Expected {n_features}, but got {W.shape[1]}.")
    if not np.all(np.isfinite(W)):
        raise ValueError("Input contains NaN or infinity.")
    if (W < 0).any():
        raise ValueError("Negative values in data passed to NMF (input H)")
    if W.max() == 0:
        raise ValueError("Array passed to NMF (input H) is full of zeros.")
    return W


def _exchange_allowed(hint):
    """hint -> solve_activations(cooperative=...).  "throughput" (default): the solver may use the kernels whose
    workgroups exchange partial sums through device memory (k_fused_all / cooperative k_fused_res: fastest, but the
    call then ends with ONE host synchronisation, to learn whether a bounded wait timed out - include/evc.h, "Host
    synchronisation").  "latency": EVC_FLAG_NO_EXCHANGE - no such kernel, the launch sequence is asynchronous up
    to the copy of the results (for callers that queue several solves on streams)."""
    if hint not in ("throughput", "latency"):
        raise ValueError(f"hint must be 'throughput' or 'latency', got {hint!r}")
    return hint == "throughput"


def _factorize(X, W, beta_loss="kullback-leibler", tol=1e-4, *, device=None, algo="auto",
               honor_beta_loss=False, hint="throughput"):
    """H (N x T) with W.T @ H ~ X.T.  X: (T, M) frames as rows, W: (N, M) exemplars as rows.

    As in the reference the Frobenius loss is forced whatever `beta_loss` says, the
    activations start at sqrt(mean(X)/N), denominators that are exactly 0 become
    1.1920929e-7, and every 10 iterations the loop stops when the Frobenius error decreased by
    less than `tol` (relative to the initial error); at most 150 iterations.
    """
    return _factorize_impl(X, W, beta_loss, tol, device, algo, honor_beta_loss, hint, False)[0]


def _factorize_impl(X, W, beta_loss, tol, device, algo, honor_beta_loss, hint, with_recon, warn_sink=None):
    """_factorize; with_recon: also H.T @ W (T x M), the reconstruction the WORLD branch's residual needs
    (04_align_n_nmf.py:292-294), formed by the same launch sequence from the device-resident activations.
    warn_sink: a list that receives (message, category) instead of warnings.warn - for callers on worker threads
    (`warnings.catch_warnings` swaps process-global state and must not be entered from several threads at once);
    the caller's thread issues them."""
    def _warn(message, category, stacklevel):
        if warn_sink is not None:
            warn_sink.append((message, category))
        else:
            warnings.warn(message, category, stacklevel=stacklevel + 1)

    # 04_align_n_nmf.py:210 overrides `beta_loss` with "frobenius" whatever the caller passed;
    # honor_beta_loss=True runs the loss that was asked for (sklearn's KL update)
    loss = beta_loss if honor_beta_loss else "frobenius"
    if loss not in ("frobenius", "kullback-leibler"):
        raise ValueError(f"Invalid beta_loss parameter: got {loss!r}")
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError(f"Expected 2D array, got {X.ndim}D array instead")
    W = _check_dictionary(W, X.shape[1])
    if X.dtype not in (np.float64, np.float32):
        X = X.astype(np.float64)
    if W.dtype != X.dtype:
        raise TypeError(f"H should have the same dtype as X. Got H.dtype = {W.dtype}.")
    if (X < 0).any():
        _warn("X has negative entries; the multiplicative update is only meaningful for "
              "non-negative data (scikit-learn does not check X on this route)", RuntimeWarning, 2)
    kw = dict(layout="frame_major", iters=MAX_ITER, eps_mode="zero_replace", init="sklearn",
              check_every=CHECK_EVERY if tol > 0 else 0, stop_rule="sklearn" if tol > 0 else "none",
              tol=tol, algo=algo, device=device, info=True, loss=loss, cooperative=_exchange_allowed(hint))
    recon = None
    if with_recon:      # Y = "B" H with the SOURCE dictionary in B's place: H never makes a round trip
        act, recon, info = _solve_and_synthesize(W, X, W, **kw)
    else:
        act, info = solve_activations(W, X, **kw)
    if tol > 0 and int(info["n_iter"][0]) == MAX_ITER:
        _warn(f"Maximum number of iterations {MAX_ITER} reached. Increase it to improve convergence.",
              ConvergenceWarning, 3)
    return act.T, recon


def factorize_utterances(X_list, W, tol=1e-4, *, device=None, algo="auto", max_iter=MAX_ITER, hint="throughput",
                         return_info=False):
    """`_factorize` for many utterances in ONE launch sequence: the frames are concatenated,
    the per-call semantics (initial value, stop test) are applied per utterance on the device.
    Returns a list of (N x T_u) arrays and the per-utterance iteration counts (return_info: also the solver's
    info dict - which kernel carried the loop, redo, ...)."""
    X_list = [np.asarray(x) for x in X_list]
    W = _check_dictionary(W, X_list[0].shape[1])
    offs = np.concatenate([[0], np.cumsum([x.shape[0] for x in X_list])]).astype(np.int32)
    X = np.concatenate(X_list, axis=0).astype(W.dtype, copy=False)
    act, info = solve_activations(
        W, X, layout="frame_major", iters=max_iter, eps_mode="zero_replace", init="sklearn",
        check_every=CHECK_EVERY if tol > 0 else 0, stop_rule="sklearn" if tol > 0 else "none",
        tol=tol, algo=algo, device=device, utt_offsets=offs, info=True, cooperative=_exchange_allowed(hint))
    out = [act[offs[i]:offs[i + 1]].T for i in range(len(X_list))]
    return (out, info["n_iter"], info) if return_info else (out, info["n_iter"])


def synthesize_rows(H, B, *, device=None):
    """np.matmul(H.T, B): H (N x T) as returned by `_factorize`, B (N x Mb) target exemplars as rows
    -> (T x Mb).  04_align_n_nmf.py:371-373,391."""
    H = np.asarray(H)
    B = np.asarray(B)
    return synthesize(B, np.ascontiguousarray(H.T), layout="frame_major", device=device)


def _stack(feats, key, transform=None):
    """`A = []; for f in feats: A.extend(f[key]); np.asarray(A)` (04_align_n_nmf.py:230-246,320-324):
    the aligned exemplar frames of every file, one after the other."""
    rows = [np.asarray(f[key]) if transform is None else transform(np.asarray(f[key])) for f in feats]
    rows = [r[:, np.newaxis] if r.ndim == 1 else r for r in rows]
    return np.concatenate(rows, axis=0)


RESIDUAL_FLOOR = 1e-10   # positive floor of the log-ratio residual (not in the reference)


def factorize(tobe_converted, src_feat, *, use_stft=True, tol=1e-4, device=None, cache_dir=None,
              cache_key="content", residual="reference", hint="throughput"):
    """`factorize(tobe_converted, src_feat)` of 04_align_n_nmf.py:218-333: stack the aligned source
    exemplars into the dictionary, solve the activations of the utterance to convert (one solve per
    feature stream), and - WORLD branch - form the reference's residual.

    use_stft mirrors the script's module-level flag (config/config:12).
      use_stft=True : tobe_converted['real'] (T x 201), src_feat[i]['real'] -> ({'H_stft': N x T}, None)
      use_stft=False: 'sp', 'ap' (T x 513) and 'f0' (T,) -> ({'H_sp','H_ap','H_f0'}, {'r_sp','r_ap','r_f0'})
    The residual is the reference's literal expression np.log(H.T @ A - conv) (:292-294), which is NaN
    wherever the reconstruction undershoots; it is reproduced, not repaired.
    residual="log_ratio" (an extension, SURVEY 8f-4: what the script's comment `log r_n = log y_n -
    log y_hat_n` (:263) describes) stores r = log max(y, floor) - log max(H.T @ A, floor) instead; pass the
    same value to convert().

    cache_dir (the reference hard-codes its working directory; None = no files): the activation /
    residual pickles of :251-260,296-308,330-331 are read when present and written after a solve
    (compat.artifacts).  The reference names them by the NUMBER of dictionary files only, so a second
    utterance gets the first one's H back; cache_key="content" (default) adds a digest of the
    utterance and the dictionary to the name, cache_key="reference" reproduces the reference's name
    (and its hazard) so that caches written by the reference are found.

    hint: "throughput" (default) or "latency" - see _exchange_allowed(): "latency" keeps the solver to kernels
    without inter-workgroup exchange, whose launch sequence needs no host synchronisation.
    """
    from . import artifacts
    if use_stft:
        conv_stft = np.abs(np.asarray(tobe_converted["real"]))
        A_stft = _stack(src_feat, "real", np.abs)
        hpath = None
        if cache_dir is not None:
            dg = artifacts.content_digest(conv_stft, A_stft) if cache_key == "content" else None
            hpath = artifacts.activation_cache_path(cache_dir, True, len(src_feat), "H", cache_key, dg)
            if os.path.isfile(hpath):
                return artifacts.read_activations(hpath, True), None
        H = {"H_stft": _factorize(conv_stft, A_stft, tol=tol, device=device, hint=hint)}
        if hpath is not None:
            artifacts.write_activations(hpath, H)
        return H, None
    streams = {"sp": (np.asarray(tobe_converted["sp"]), _stack(src_feat, "sp")),
               "ap": (np.asarray(tobe_converted["ap"]), _stack(src_feat, "ap")),
               "f0": (np.asarray(tobe_converted["f0"])[:, np.newaxis], _stack(src_feat, "f0"))}
    hpath = rpath = None
    H, R = {}, {}
    if cache_dir is not None:
        dg = (artifacts.content_digest(*[a for pair in streams.values() for a in pair])
              if cache_key == "content" else None)
        hpath = artifacts.activation_cache_path(cache_dir, False, len(src_feat), "H", cache_key, dg)
        rpath = artifacts.activation_cache_path(cache_dir, False, len(src_feat), "R", cache_key, dg)
        if os.path.isfile(hpath):
            H = artifacts.read_activations(hpath, False)
            if os.path.isfile(rpath):
                return H, artifacts.read_residuals(rpath)
    if residual not in ("reference", "log_ratio"):
        raise ValueError("residual must be 'reference' or 'log_ratio'")
    # The three streams are independent solves: each runs on a stream of its own, issued from a host thread of its
    # own (one C3-sized solve fills about half of the chip), and its reconstruction H.T @ A comes out of the same
    # launch sequence - the activations are downloaded once and never uploaded again.
    recons = _solve_streams({n: v for n, v in streams.items() if "H_" + n not in H}, tol, device, hint, H)
    for name, (conv, A) in streams.items():
        # a cached H without its R: the residual is recomputed from it (:261-276; the expression there,
        # np.matmul(A, H), has its operands the wrong way round and cannot run - :292-294's is used)
        recon = recons[name] if name in recons else synthesize_rows(H["H_" + name], A, device=device)
        if residual == "log_ratio":
            R["r_" + name] = np.log(np.maximum(conv, RESIDUAL_FLOOR)) - np.log(np.maximum(recon, RESIDUAL_FLOOR))
        else:
            with np.errstate(invalid="ignore", divide="ignore"):
                R["r_" + name] = np.log(recon - conv)
    if hpath is not None:
        if not os.path.isfile(hpath):
            artifacts.write_activations(hpath, H)
        artifacts.write_residuals(rpath, R)
    return H, R


_side_streams = {}      # (device index, slot) -> torch.cuda.Stream: reused, so that the solver's per-stream scratch is too


def _factorize_recon(conv, A, tol, device, hint, warn_sink=None):
    """(H, H.T @ A) of one stream: `_factorize` plus the reconstruction, in one launch sequence (a seam of its own so
    that the host-logic tests can put the oracle in its place)"""
    return _factorize_impl(conv, A, "frobenius", tol, device, "auto", False, hint, True, warn_sink)


def _solve_streams(streams, tol, device, hint, H_out):
    """solve every (conv, A) of `streams` concurrently: one host thread and one HIP stream each (the C ABI is
    thread-safe for distinct streams, include/evc.h); fills H_out['H_<name>'] and returns {name: H.T @ A}.
    The solves' warnings (ConvergenceWarning, negative X) are collected per stream and issued by the caller's thread
    after the join - no thread enters `warnings.catch_warnings`, whose save / restore of the process-global filter list
    is only correct when the exits are last-in-first-out; the first exception is re-raised."""
    import threading
    import torch
    from ..solver import require_device
    if not streams:
        return {}
    recons, caught, errors = {}, {}, {}
    if not torch.cuda.is_available():      # nothing to overlap: the solve itself raises (there is no CPU fallback)
        for n, (conv, A) in streams.items():
            H_out["H_" + n], recons[n] = _factorize_recon(conv, A, tol, device, hint)
        return recons
    dev = require_device(device)

    def work(name, conv, A, slot):
        try:
            torch.cuda.set_device(dev)
            st = _side_streams.get((dev.index, slot))
            if st is None:
                st = _side_streams[(dev.index, slot)] = torch.cuda.Stream(device=dev)
            sink = []
            with torch.cuda.stream(st):
                h, r = _factorize_recon(conv, A, tol, dev, hint, sink)
            H_out["H_" + name], recons[name], caught[name] = h, r, sink
        except BaseException as e:  # noqa: BLE001 - handed to the caller's thread
            errors[name] = e

    names = list(streams)
    threads = [threading.Thread(target=work, args=(n, *streams[n], k + 1)) for k, n in enumerate(names[1:])]
    for t in threads:
        t.start()
    work(names[0], *streams[names[0]], 0)
    for t in threads:
        t.join()
    for n in names:
        if n in errors:
            raise errors[n]
        for message, category in caught.get(n, ()):
            warnings.warn(message, category, stacklevel=3)
    return recons


def convert(H, tar_feat, residual=None, *, use_stft=True, device=None, residual_mode="reference"):
    """`convert(H, tar_feat, residual)` of 04_align_n_nmf.py:336-393: stack the parallel target
    exemplars and synthesise H.T @ B per stream; the WORLD branch applies the reference's residual
    expression exp(log(H.T @ B) + log(r)) with NaNs of r zeroed first (:367-373), literally;
    residual_mode="log_ratio" applies the residual of factorize(residual="log_ratio"):
    exp(log max(H.T @ B, floor) + r)."""
    if use_stft:
        return synthesize_rows(H["H_stft"], _stack(tar_feat, "real", np.abs), device=device)
    out = {}
    for name in ("sp", "ap", "f0"):
        B = _stack(tar_feat, name)
        if residual_mode == "log_ratio":
            y = np.maximum(synthesize_rows(H["H_" + name], B, device=device), RESIDUAL_FLOOR)
            out[name] = np.exp(np.log(y) + np.asarray(residual["r_" + name]))
            continue
        if residual_mode != "reference":
            raise ValueError("residual_mode must be 'reference' or 'log_ratio'")
        r = np.array(residual["r_" + name], copy=True)
        r[np.isnan(r)] = 0
        with np.errstate(invalid="ignore", divide="ignore"):
            out[name] = np.exp(np.log(synthesize_rows(H["H_" + name], B, device=device)) + np.log(r))
    out["f0"] = np.squeeze(out["f0"])
    return out
