"""S2: pymf's `NMF(data, num_bases).factorize(compute_w=False)` on the GPU.

Mirrors /root/reference/dependencies/pymf-29e3490d.../pymf/nmf.py:21-77 and
pymf/base.py:102-270: `data` is M x T, `W` M x N (set by the caller), `H` N x T is created
lazily as random((N,T)) + 1e-4 from the global numpy RNG, updated IN PLACE by
`factorize`, with `ferr[i] = ||data - W H||_F` per iteration and the machine-epsilon stop.
Only the fixed-dictionary path (`compute_w=False`) is the accelerated path.  pymf's DEFAULT is
`compute_w=True` (base.py:208): the dictionary update (nmf.py:72-76, a few small products per
iteration) then runs on the host in numpy, as SURVEY 8(b) S2 allows, while every activation
update still goes through the GPU solver - a drop-in must not raise on the default arguments
(round 3 did).  A RuntimeWarning says so once per call.
"""
from __future__ import annotations

import logging

import numpy as np

from ..solver import frame_residuals, solve_activations, synthesize

_EPS = np.finfo(float).eps


class NMF:
    _EPS = _EPS

    def __init__(self, data, num_bases=4, **kwargs):
        self._logger = logging.getLogger("pymf")
        self.data = data
        self._num_bases = num_bases
        self._data_dimension, self._num_samples = self.data.shape
        self._device = kwargs.get("device")
        self._algo = kwargs.get("algo", "auto")

    # -- pymf/base.py:133-165 --
    def residual(self):
        WH = synthesize(np.asarray(self.W, dtype=np.float64), np.asarray(self.H, dtype=np.float64),
                        layout="bin_major", device=self._device)
        res = np.sum(np.abs(self.data - WH))
        return 100.0 * res / np.sum(np.abs(self.data))

    def frobenius_norm(self):
        if hasattr(self, "H") and hasattr(self, "W"):
            e2 = frame_residuals(self.W, self.data, self.H, layout="bin_major", dtype="f64",
                                 device=self._device)
            return np.sqrt(np.sum(e2))
        return None

    def _init_h(self):
        self.H = np.random.random((self._num_bases, self._num_samples)) + 10 ** -4

    def _init_w(self):
        self.W = np.random.random((self._data_dimension, self._num_bases)) + 10 ** -4

    def _update_w(self):
        """pymf/nmf.py:72-76: W <- W (.) (data H^T) (/) (W H H^T + 1e-9), columns scaled to unit norm; in place.
        Host numpy: the dictionary update is not on the accelerated path."""
        W2 = np.dot(np.dot(self.W, self.H), self.H.T) + 10 ** -9
        self.W *= np.dot(self.data, self.H.T)
        self.W /= W2
        self.W /= np.sqrt(np.sum(self.W ** 2.0, axis=0))

    def _factorize_with_dictionary_update(self, niter, compute_h, compute_err):
        """pymf/base.py:238-270 with compute_w=True: per iteration W (host), then H (one update on the GPU), then the
        error and the machine-epsilon stop test"""
        import warnings
        warnings.warn("pymf's dictionary update (compute_w=True) is outside the accelerated path: it runs in numpy on "
                      "the host, one GPU activation update per iteration; pass compute_w=False for the fixed-dictionary "
                      "solve", RuntimeWarning, stacklevel=3)
        if not hasattr(self, "W"):
            self._init_w()
        if not hasattr(self, "H") and compute_h:
            self._init_h()
        self.W = np.asarray(self.W, dtype=np.float64)
        if compute_err:
            self.ferr = np.zeros(niter)
        for i in range(niter):
            self._update_w()
            if compute_h:
                self.H[...] = solve_activations(
                    self.W, np.asarray(self.data, dtype=np.float64), np.asarray(self.H, dtype=np.float64),
                    layout="bin_major", iters=1, eps_mode="add", eps=10 ** -9, init="given", algo=self._algo,
                    device=self._device)
            if compute_err:
                self.ferr[i] = self.frobenius_norm()
                self._logger.info("FN: %s (%s/%s)" % (self.ferr[i], i + 1, niter))
                if i > 1 and np.abs(self.ferr[i] - self.ferr[i - 1]) / self._num_samples < self._EPS:
                    self.ferr = self.ferr[:i]
                    break
            else:
                self._logger.info("Iteration: (%s/%s)" % (i + 1, niter))

    def factorize(self, niter=100, show_progress=False, compute_w=True, compute_h=True,
                  compute_err=True):
        if show_progress:
            self._logger.setLevel(logging.INFO)
        else:
            self._logger.setLevel(logging.ERROR)
        if compute_w:
            return self._factorize_with_dictionary_update(niter, compute_h, compute_err)
        if not hasattr(self, "W"):
            raise AttributeError("set .W (data_dimension x num_bases) before factorize(compute_w=False)")
        if not hasattr(self, "H") and compute_h:
            self._init_h()
        if compute_err:
            self.ferr = np.zeros(niter)
        if not compute_h or niter <= 0:
            return
        H, info = solve_activations(
            np.asarray(self.W, dtype=np.float64), np.asarray(self.data, dtype=np.float64),
            np.asarray(self.H, dtype=np.float64), layout="bin_major", iters=niter,
            eps_mode="add", eps=10 ** -9, init="given", algo=self._algo,
            check_every=1 if compute_err else 0, stop_rule="pymf" if compute_err else "none",
            tol=self._EPS, device=self._device, info=True)
        self.H[...] = H                     # in place, like `self.H *= ...; self.H /= ...`
        if compute_err:
            n = int(info["n_iter"][0])      # updates applied
            ferr = info["err"][0, 1:1 + n]
            stopped = n < niter or (n >= 3 and abs(ferr[n - 1] - ferr[n - 2]) / self._num_samples < self._EPS)
            if stopped:
                self.ferr = ferr[:n - 1].copy()   # base.py:268: ferr = ferr[:i], i = n-1
            else:
                self.ferr[:n] = ferr
            for i, e in enumerate(self.ferr):
                self._logger.info("FN: %s (%s/%s)" % (e, i + 1, niter))
