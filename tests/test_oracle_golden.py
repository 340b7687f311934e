"""The oracle (numpy restatement) against the committed golden vectors - CPU only.

The vectors were produced by the reference's own arithmetic (tools/make_golden.py):
scikit-learn 1.7.2 through the exact call of 04_align_n_nmf.py:212-213, and the vendored pymf.
The oracle must reproduce them BIT FOR BIT; that is what licenses it as the checker of the
HIP path on inputs the fixtures do not cover.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_files, load_golden
from oracle import evc_oracle as o


@pytest.mark.parametrize("path", golden_files("sklearn_"), ids=os.path.basename)
def test_sklearn_restatement_is_bit_exact(path):
    g = load_golden(path)
    act, n_iter, trace = o.sklearn_mu_fixed_dictionary(g["X_rows"], g["W_rows"], int(g["max_iter"]),
                                                       float(g["tol"]), l1_reg=float(g["l1_reg"]))
    assert n_iter == int(g["n_iter"])
    assert np.array_equal(act.T, g["H"])
    assert np.array_equal(o.s4_convert(act.T, g["B_rows"]), g["Y_rows"])


@pytest.mark.parametrize("path", golden_files("sklearnkl_"), ids=os.path.basename)
def test_sklearn_kl_restatement_is_bit_exact(path):
    """the Kullback-Leibler default of `_factorize`'s signature, through scikit-learn."""
    g = load_golden(path)
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary_kl(g["X_rows"], g["W_rows"], int(g["max_iter"]), float(g["tol"]))
    assert n_iter == int(g["n_iter"])
    assert np.array_equal(act.T, g["H"])


def test_s1_factorize_matches_live_call_defaults():
    g = load_golden([p for p in golden_files("sklearn_") if p.endswith("m201_n128_t40_tol.npz")][0])
    H = o.s1_factorize(g["X_rows"], g["W_rows"], beta_loss="kullback-leibler", tol=1e-4)
    assert H.shape == (128, 40) and np.array_equal(H, g["H"])


@pytest.mark.parametrize("path", golden_files("pymf_"), ids=os.path.basename)
def test_pymf_restatement_is_bit_exact(path):
    g = load_golden(path)
    H, ferr = o.pymf_factorize(g["data"], g["W"], g["H0"], int(g["niter"]), bool(g["compute_err"]))
    assert np.array_equal(H, g["H"])
    if bool(g["compute_err"]):
        assert np.array_equal(ferr, g["ferr"])
        np.testing.assert_allclose(o.pymf_frobenius(g["data"], g["W"], H), float(g["frobenius_norm"]), rtol=1e-15)


@pytest.mark.parametrize("path", golden_files("pymfw_"), ids=os.path.basename)
def test_pymf_default_call_restatement_is_bit_exact(path):
    """factorize() with pymf's defaults (compute_w=True): the vendored pymf's W, H and ferr, bit for bit"""
    g = load_golden(path)
    W, H, ferr = o.pymf_factorize_full(g["data"], g["W0"], g["H0"], int(g["niter"]), bool(g["compute_err"]))
    assert np.array_equal(W, g["W"]) and np.array_equal(H, g["H"])
    if bool(g["compute_err"]):
        assert np.array_equal(ferr, g["ferr"])


def test_pymf_doctest_known_answer():
    """pymf/nmf.py:57-63 - the reference's only known answer on the fixed-dictionary path."""
    data = np.array([[1.5], [1.2]])
    W = np.array([[1.0, 0.0], [0.0, 1.0]])
    rng = np.random.default_rng(0)
    H, ferr = o.pymf_factorize(data, W, rng.random((2, 1)) + 1e-4, niter=20, compute_err=True)
    np.testing.assert_allclose(H, data, rtol=1e-8)
    assert len(ferr) == 2          # loop leaves at i == 2, ferr truncated to [:2]


@pytest.mark.parametrize("mode,eps", [(o.EPS_ADD, 1e-9), (o.EPS_ZERO_REPLACE, o.SK_EPSILON),
                                      (o.EPS_NONE, 0.0), (o.EPS_CLAMP, 1e-15)])
def test_three_algebras_agree(mode, eps):
    """gram / literal / factored differ by rounding only (this is what lets the HIP fast path
    re-associate A^T (A H))."""
    p = o.synth_problem(25, 96, 40, seed=5)
    H0 = np.random.default_rng(1).random((96, 40)) + 1e-4
    ref = o.mu_solve(p["A"], p["X"], H0, 60, eps_mode=mode, eps=eps, algo="gram")
    for algo in ("literal", "factored"):
        got = o.mu_solve(p["A"], p["X"], H0, 60, eps_mode=mode, eps=eps, algo=algo)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=0)


def test_generic_statement_matches_surfaces():
    p = o.synth_problem(25, 64, 32, seed=101)
    X_rows, W_rows = np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T)
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary(X_rows, W_rows, 50, 0.0)
    H0 = np.full((64, 32), o.sklearn_init_value(X_rows, 64))
    got = o.mu_solve(p["A"], p["X"], H0, 50, eps_mode=o.EPS_ZERO_REPLACE, eps=o.SK_EPSILON, algo="gram")
    np.testing.assert_allclose(got, act.T, rtol=1e-11)
    H0 = np.random.default_rng(2).random((64, 32)) + 1e-4
    Hp, _ = o.pymf_factorize(p["X"], p["A"], H0, 30, compute_err=False)
    got = o.mu_solve(p["A"], p["X"], H0, 30, eps_mode=o.EPS_ADD, eps=1e-9, algo="literal")
    np.testing.assert_allclose(got, Hp, rtol=1e-12)


def test_mu_monotone_and_nonnegative():
    p = o.synth_problem(40, 48, 33, seed=9)
    H = np.random.default_rng(3).random((48, 33)) + 1e-4
    prev = o.residual_fro(p["A"], p["X"], H)
    for _ in range(20):
        H = o.mu_solve(p["A"], p["X"], H, 1)
        cur = o.residual_fro(p["A"], p["X"], H)
        assert cur <= prev * (1 + 1e-12) and (H >= 0).all()
        prev = cur


@pytest.mark.parametrize("path", golden_files("gl_"), ids=os.path.basename)
def test_griffin_lim_restatement_is_bit_exact(path):
    """vectors produced by the reference's own zz_audio_utilities.py."""
    g = load_golden(path)
    x, rmse = o.griffin_lim(g["mag"], int(g["n_fft"]), int(g["hop"]), int(g["iters"]), g["x0"])
    assert np.array_equal(x, g["x"])
    np.testing.assert_allclose(rmse, g["rmse"], rtol=1e-12)     # the reference prints repr-rounded floats


def test_oracle_reproduces_the_dictionary_scale_audio_fixture():
    """tests/golden/audio_stft_n4096_f64.npz (tools/make_golden_audio.py: installed scikit-learn, N = 4096, T = 688,
    real audio): bit for bit.  The stop rule couples all frames of a call, so the whole utterance is run (about ten
    seconds on 8 cores)."""
    from oracle import evc_oracle as o
    g32 = load_golden(os.path.join(GOLDEN, "audio_stft_n4096_f32.npz"))
    g = load_golden(os.path.join(GOLDEN, "audio_stft_n4096_f64.npz"))
    A, B, X = (g32[k].astype(np.float64) for k in ("A_rows", "B_rows", "X_rows"))
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary(X, A, int(g["max_iter"]), float(g["tol"]))
    assert n_iter == int(g["n_iter"]) == 140
    H = act.T          # the oracle returns scikit-learn's W (T x N); the script returns its transpose
    assert np.array_equal(H[:, :32], g["H_first32"])
    assert np.array_equal(o.s4_convert(H, B), g["Y_rows"])
