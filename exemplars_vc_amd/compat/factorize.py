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

import warnings

import numpy as np

from ..solver import solve_activations, synthesize

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


def _factorize(X, W, beta_loss="kullback-leibler", tol=1e-4, *, device=None, algo="auto",
               honor_beta_loss=False):
    """H (N x T) with W.T @ H ~ X.T.  X: (T, M) frames as rows, W: (N, M) exemplars as rows.

    As in the reference the Frobenius loss is forced whatever `beta_loss` says, the
    activations start at sqrt(mean(X)/N), denominators that are exactly 0 become
    1.1920929e-7, and every 10 iterations the loop stops when the Frobenius error decreased by
    less than `tol` (relative to the initial error); at most 150 iterations.
    """
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
        warnings.warn("X has negative entries; the multiplicative update is only meaningful for "
                      "non-negative data (scikit-learn does not check X on this route)",
                      RuntimeWarning, stacklevel=2)
    act, info = solve_activations(
        W, X, layout="frame_major", iters=MAX_ITER, eps_mode="zero_replace", init="sklearn",
        check_every=CHECK_EVERY if tol > 0 else 0, stop_rule="sklearn" if tol > 0 else "none",
        tol=tol, algo=algo, device=device, info=True, loss=loss)
    if tol > 0 and int(info["n_iter"][0]) == MAX_ITER:
        warnings.warn(f"Maximum number of iterations {MAX_ITER} reached. Increase it to improve "
                      "convergence.", ConvergenceWarning, stacklevel=2)
    return act.T


def factorize_utterances(X_list, W, tol=1e-4, *, device=None, algo="auto", max_iter=MAX_ITER):
    """`_factorize` for many utterances in ONE launch sequence: the frames are concatenated,
    the per-call semantics (initial value, stop test) are applied per utterance on the device.
    Returns a list of (N x T_u) arrays and the per-utterance iteration counts."""
    X_list = [np.asarray(x) for x in X_list]
    W = _check_dictionary(W, X_list[0].shape[1])
    offs = np.concatenate([[0], np.cumsum([x.shape[0] for x in X_list])]).astype(np.int32)
    X = np.concatenate(X_list, axis=0).astype(W.dtype, copy=False)
    act, info = solve_activations(
        W, X, layout="frame_major", iters=max_iter, eps_mode="zero_replace", init="sklearn",
        check_every=CHECK_EVERY if tol > 0 else 0, stop_rule="sklearn" if tol > 0 else "none",
        tol=tol, algo=algo, device=device, utt_offsets=offs, info=True)
    return [act[offs[i]:offs[i + 1]].T for i in range(len(X_list))], info["n_iter"]


def convert(H, B, *, device=None):
    """Converted spectrogram np.matmul(H.T, B): H (N x T) as returned by `_factorize`,
    B (N x Mb) target exemplars as rows -> (T x Mb).  04_align_n_nmf.py:391."""
    H = np.asarray(H)
    B = np.asarray(B)
    return synthesize(B, np.ascontiguousarray(H.T), layout="frame_major", device=device)
