#!/usr/bin/env python3
"""Generate tests/golden/*.npz - runs ONLY in the build container.

The vectors are produced by executing the reference's own arithmetic here:

  * scikit-learn 1.7.2 (installed in the image) through exactly the call that
    /root/reference/04_align_n_nmf.py:212-213 makes
    (`non_negative_factorization(X=X, H=W, init="custom", update_H=False,
    n_components=W.shape[0], beta_loss="frobenius", solver="mu", tol=tol,
    max_iter=150)`), plus the `alpha_W`/`l1_ratio` hook for the L1 variant.
  * the vendored pymf (`/root/reference/dependencies/pymf-29e3490.../pymf`)
    imported from where it lies.  Its module header imports two symbols that
    the hot path never touches and that no longer exist in this image
    (`scipy.misc.factorial`, moved to `scipy.special`; `cvxopt`, used only by
    NMFALS/NMFNNLS).  For the import to succeed this script - and only this
    script - aliases `scipy.misc.factorial` to `scipy.special.factorial` and
    registers an empty placeholder module named `cvxopt`.  `NMF._update_h`,
    `PyMFBase.factorize`, `_init_h`, `_converged` and `frobenius_norm` run
    unmodified.  If the import fails the pymf fixtures are skipped and the
    oracle stays pinned for that surface by pymf's doctest known answer only.
  * real audio: |Re(STFT)| (n_fft=400, hop=80, hann - 04_align_n_nmf.py:46-48,
    422-427) of three of the reference's sample wavs, restated in numpy
    because librosa is absent.  Only the derived magnitudes (data) are stored.

Nothing of the reference's source text is written into the fixtures: each
.npz holds inputs, expected outputs and a few scalars.
"""
import os
import sys
import types
import wave
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
PYMF_DIR = os.path.join(REF, "dependencies", "pymf-29e3490d0020a656176834bc195c5906bc063419")


def synth(M, N, T, seed, active=8, Mb=None):
    rng = np.random.default_rng(seed)
    Mb = M if Mb is None else Mb
    A = rng.random((M, N)) + 1e-3
    A /= np.linalg.norm(A, axis=0, keepdims=True)
    B = rng.random((Mb, N)) + 1e-3
    B /= np.linalg.norm(B, axis=0, keepdims=True)
    Hs = rng.random((N, T)) * (rng.random((N, T)) < (active / N))
    X = A @ Hs + 1e-6
    return A, B, X


def run_sklearn(X_rows, W_rows, tol, max_iter=150, alpha_W=0.0, l1_ratio=0.0, beta_loss="frobenius"):
    from sklearn.decomposition import non_negative_factorization
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _W, _H, n_iter = non_negative_factorization(
            X=X_rows, H=W_rows, init="custom", update_H=False, n_components=W_rows.shape[0],
            beta_loss=beta_loss, solver="mu", tol=tol, max_iter=max_iter, verbose=0,
            alpha_W=alpha_W, l1_ratio=l1_ratio)
    return _W, n_iter


def sklearn_case(name, M, N, T, seed, tol, max_iter=150, alpha_W=0.0, l1_ratio=0.0,
                 zero_cols=0, A=None, B=None, X=None, beta_loss="frobenius"):
    if A is None:
        A, B, X = synth(M, N, T, seed)
    if zero_cols:
        # all-zero frames: numerator 0 -> activations hit exact 0; exercises the
        # `denominator == 0 -> EPSILON` rule only through H=0 rows downstream.
        X = X.copy()
        X[:, :zero_cols] = 0.0
    X_rows = np.ascontiguousarray(X.T)      # T x M  (script orientation)
    W_rows = np.ascontiguousarray(A.T)      # N x M
    B_rows = np.ascontiguousarray(B.T)      # N x Mb
    act, n_iter = run_sklearn(X_rows, W_rows, tol, max_iter, alpha_W, l1_ratio, beta_loss)
    H = act.T                               # what _factorize returns (N x T)
    Y_rows = np.matmul(H.T, B_rows)         # convert(): T x Mb
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        surface="sklearn", beta_loss=beta_loss, X_rows=X_rows, W_rows=W_rows, B_rows=B_rows, H=np.ascontiguousarray(H),
        Y_rows=Y_rows, n_iter=n_iter, tol=tol, max_iter=max_iter,
        l1_reg=X_rows.shape[1] * alpha_W * l1_ratio, alpha_W=alpha_W, l1_ratio=l1_ratio)
    print(f"{name}: T={X_rows.shape[0]} M={X_rows.shape[1]} N={W_rows.shape[0]} n_iter={n_iter}")


def import_pymf():
    import scipy.misc
    import scipy.special
    if not hasattr(scipy.misc, "factorial"):
        scipy.misc.factorial = scipy.special.factorial
    if "cvxopt" not in sys.modules:
        ph = types.ModuleType("cvxopt")
        ph.solvers = types.ModuleType("cvxopt.solvers")
        ph.base = types.ModuleType("cvxopt.base")
        sys.modules["cvxopt"] = ph
        sys.modules["cvxopt.solvers"] = ph.solvers
        sys.modules["cvxopt.base"] = ph.base
    sys.path.insert(0, PYMF_DIR)
    from pymf.nmf import NMF
    return NMF


def pymf_case(NMF, name, M, N, T, seed, niter, compute_err, data=None, W=None, H0=None):
    if data is None:
        A, B, X = synth(M, N, T, seed)
        data, W = X, A
        rng = np.random.default_rng(seed + 1)
        H0 = rng.random((N, T)) + 1e-4
    else:
        B = np.eye(W.shape[0], W.shape[1])
    mdl = NMF(data.copy(), num_bases=W.shape[1])
    mdl.W = W.copy()
    if H0 is not None:
        mdl.H = H0.copy()
    else:
        np.random.seed(1234)
        st = np.random.get_state()
        np.random.set_state(st)
        H0 = np.random.random((W.shape[1], data.shape[1])) + 1e-4   # what _init_h will draw
        np.random.set_state(st)
    mdl.factorize(niter=niter, compute_w=False, compute_err=compute_err)
    ferr = np.asarray(mdl.ferr) if compute_err else np.zeros(0)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        surface="pymf", data=data, W=W, B=B, H0=H0, H=mdl.H, Y=B @ mdl.H, niter=niter,
        compute_err=compute_err, ferr=ferr,
        frobenius_norm=mdl.frobenius_norm(), residual=mdl.residual())
    print(f"{name}: M={data.shape[0]} T={data.shape[1]} N={W.shape[1]} len(ferr)={len(ferr)}")


def pymf_full_case(NMF, name, M, N, T, seed, niter, compute_err):
    """pymf's DEFAULT call, factorize(compute_w=True): dictionary and activations updated in turn (base.py:238-270,
    nmf.py:66-76).  Outside the accelerated path; the fixture pins compat.pymf's host-side dictionary update."""
    A, B, X = synth(M, N, T, seed)
    rng = np.random.default_rng(seed + 1)
    W0 = rng.random((M, N)) + 1e-4
    H0 = rng.random((N, T)) + 1e-4
    mdl = NMF(X.copy(), num_bases=N)
    mdl.W = W0.copy()
    mdl.H = H0.copy()
    mdl.factorize(niter=niter, compute_w=True, compute_err=compute_err)
    ferr = np.asarray(mdl.ferr) if compute_err else np.zeros(0)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), surface="pymf_full", data=X, W0=W0, H0=H0, W=mdl.W, H=mdl.H,
                        niter=niter, compute_err=compute_err, ferr=ferr)
    print(f"{name}: M={M} T={T} N={N} len(ferr)={len(ferr)}")


def read_wav(path):
    with wave.open(path) as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return pcm.astype(np.float64) / 32768.0


def stft_abs_real(y, n_fft=400, hop=80):
    """|Re(STFT)| frames-as-rows, centre-padded (reflect) with a periodic hann
    window - the librosa.stft defaults 04_align_n_nmf.py:422 relies on."""
    y = np.pad(y, n_fft // 2, mode="reflect")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.abs(np.real(np.fft.rfft(y[idx] * win, axis=1)))    # T x 201


def audio_case():
    src = stft_abs_real(read_wav(os.path.join(REF, "data/SF1/100001.wav")))
    tar = stft_abs_real(read_wav(os.path.join(REF, "data/TF1/100001.wav")))
    conv = stft_abs_real(read_wav(os.path.join(REF, "data/SF1/100002.wav")))
    n = min(len(src), len(tar))
    # exemplars: the 160 most energetic source frames and the target frames at the same
    # indices (a stand-in for DTW alignment, which is outside the path); 40 frames to convert.
    order = np.sort(np.argsort(-src[:n].sum(1))[:160])
    A = src[order].T + 1e-9
    B = tar[order].T + 1e-9
    X = conv[100:140].T
    sklearn_case("sklearn_audio_stft", 201, 160, 40, 0, tol=1e-4, A=A, B=B, X=X)


def griffin_lim_cases():
    """the reference's own Griffin-Lim (zz_audio_utilities.py, numpy + `from pylab
    import *`; matplotlib is installed here) on magnitude spectrograms of its sample audio."""
    import contextlib
    import io
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import zz_audio_utilities as zz
    pcm = read_wav(os.path.join(REF, "data/SF1/100002.wav"))
    for name, sl, n_fft, hop, iters, seed in [("gl_t96_fft400_hop80_k25", slice(60, 156), 400, 80, 25, 11),
                                              ("gl_t50_fft256_hop64_k40", slice(100, 150), 256, 64, 40, 12),
                                              ("gl_t33_fft400_hop80_k3", slice(10, 43), 400, 80, 3, 13)]:
        mag = np.abs(zz.stft_for_reconstruction(pcm, n_fft, hop))[sl]
        np.random.seed(seed)
        st = np.random.get_state()
        x0 = np.random.randn(int(mag.shape[0] * hop + n_fft))        # what the reference will draw
        np.random.set_state(st)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            x = zz.reconstruct_signal_griffin_lim(mag, n_fft, hop, iters)
        rmse = np.array([float(l.split("RMSE:")[1]) for l in out.getvalue().splitlines() if "RMSE" in l])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), surface="griffin_lim", mag=mag, n_fft=n_fft, hop=hop,
                            iters=iters, x0=x0, x=x, rmse=rmse)
        print(f"{name}: T={mag.shape[0]} bins={mag.shape[1]} len={len(x)} final rmse={rmse[-1]:.3e}")


def main():
    os.makedirs(OUT, exist_ok=True)
    griffin_lim_cases()
    # G1: the live call, fixed 50-iteration budget and the script's default (tol=1e-4, <=150)
    sklearn_case("sklearn_m25_n64_t32_k50", 25, 64, 32, 101, tol=0.0, max_iter=50)
    sklearn_case("sklearn_m201_n128_t40_tol", 201, 128, 40, 102, tol=1e-4)
    sklearn_case("sklearn_m201_n128_t40_tol1e-2", 201, 128, 40, 102, tol=1e-2)  # early stop (n_iter=130)
    sklearn_case("sklearn_m25_n64_t50_tol5e-2", 25, 64, 50, 107, tol=5e-2)      # early stop
    sklearn_case("sklearn_m513_n96_t21_tol", 513, 96, 21, 103, tol=1e-4)     # WORLD sp width, ragged T
    sklearn_case("sklearn_m1_n48_t37_tol", 1, 48, 37, 104, tol=1e-4)         # the f0 stream (M=1)
    sklearn_case("sklearn_zero_frames", 25, 64, 24, 105, tol=0.0, max_iter=30, zero_cols=3)
    # G2: L1-penalised MU through sklearn's alpha_W / l1_ratio hook
    sklearn_case("sklearn_l1_m25_n256_t64_k100", 25, 256, 64, 106, tol=0.0, max_iter=100,
                 alpha_W=0.01, l1_ratio=1.0)
    audio_case()
    # the KL default of _factorize's signature (overridden by the script at :210)
    kl = "kullback-leibler"
    sklearn_case("sklearnkl_m25_n64_t32_k50", 25, 64, 32, 301, tol=0.0, max_iter=50, beta_loss=kl)
    sklearn_case("sklearnkl_m201_n128_t40_tol", 201, 128, 40, 302, tol=1e-4, beta_loss=kl)
    sklearn_case("sklearnkl_m25_n64_t50_tol2e-2", 25, 64, 50, 303, tol=2e-2, beta_loss=kl)
    sklearn_case("sklearnkl_zero_frames", 25, 64, 24, 304, tol=0.0, max_iter=30, zero_cols=3, beta_loss=kl)
    # G3/G4: pymf
    try:
        NMF = import_pymf()
    except Exception as e:  # noqa: BLE001 - report and continue without pymf fixtures
        print("pymf import failed, pymf fixtures skipped:", repr(e))
        return
    pymf_case(NMF, "pymf_m25_n64_t32_k50_noerr", 25, 64, 32, 201, niter=50, compute_err=False)
    pymf_case(NMF, "pymf_m25_n64_t32_k50_err", 25, 64, 32, 201, niter=50, compute_err=True)
    pymf_case(NMF, "pymf_m40_n24_t19_k400_err", 40, 24, 19, 202, niter=400, compute_err=True)
    pymf_case(NMF, "pymf_doctest_kat", 0, 0, 0, 0, niter=20, compute_err=True,
              data=np.array([[1.5], [1.2]]), W=np.array([[1.0, 0.0], [0.0, 1.0]]), H0=None)
    pymf_full_case(NMF, "pymfw_m25_n12_t40_k30_err", 25, 12, 40, 401, niter=30, compute_err=True)
    pymf_full_case(NMF, "pymfw_m25_n12_t40_k30_noerr", 25, 12, 40, 401, niter=30, compute_err=False)


if __name__ == "__main__":
    main()
