"""Parity of the HIP path (through the C ABI) with the reference arithmetic.  `-m gpu`.

Every comparison is against (i) the committed golden vectors, which were produced by the
reference's own code (scikit-learn's MU as called by 04_align_n_nmf.py, and the vendored
pymf), or (ii) the numpy oracle on seeded inputs, which is itself bit-exact against those
vectors (tests/test_oracle_golden.py).

Tolerances: the reference is float64; north_star asks for rtol 1e-4 after the same number
of iterations.  The float64 HIP path differs from numpy only in summation order, so it is
held to RTOL64 = 1e-8 pure relative error (no absolute floor) - four orders tighter than
required.  float32 (the nmf_tool surface's native type) is compared with rtol 2e-3 plus
an absolute floor of 1e-6*max|H|, because float32 MU trajectories of different GEMM
summation orders drift apart at ~K*sqrt(N)*6e-8.
"""
import os
import warnings

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu

RTOL64 = 1e-8
ALGOS = ["factored", "gram", "literal"]
# "factored" takes the fused persistent kernel when M <= 32 (float64); these variants force the
# generic two-GEMM factored path and the other frame-tile factors of the fused kernel
VARIANTS = {"factored": {}, "gram": {}, "literal": {},
            "factored_generic": {"algo": "factored", "fused": False},
            "fused_c1": {"algo": "factored", "fused_c": 1},     # general streamed kernel, 16 frames/WG
            "fused_c2": {"algo": "factored", "fused_c": 2}}     # general streamed kernel, 32 frames/WG


def variant_kw(name):
    kw = {"algo": name}
    kw.update(VARIANTS[name])
    return kw


def oracle():
    from oracle import evc_oracle
    return evc_oracle


def assert_close64(got, want, what, rtol=RTOL64):
    r, z = rel_err(got, want)
    assert r <= rtol and z == 0.0, f"{what}: max rel err {r:.3e}, max |got| where want==0 {z:.3e}"


# ----------------------------------------------------------------------------------------
# S1: _factorize / convert against scikit-learn golden vectors
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("sklearn_"), ids=os.path.basename)
@pytest.mark.parametrize("algo", list(VARIANTS))
def test_sklearn_golden_solver(path, algo):
    import exemplars_vc_amd as evc
    g = load_golden(path)
    tol, max_iter, l1 = float(g["tol"]), int(g["max_iter"]), float(g["l1_reg"])
    act, info = evc.solve_activations(
        g["W_rows"], g["X_rows"], layout="frame_major", iters=max_iter, eps_mode="zero_replace",
        init="sklearn", l1=l1, check_every=10 if tol > 0 else 0,
        stop_rule="sklearn" if tol > 0 else "none", tol=tol, info=True, **variant_kw(algo))
    assert int(info["n_iter"][0]) == int(g["n_iter"]), (info["n_iter"], g["n_iter"])
    assert_close64(act.T, g["H"], f"H {os.path.basename(path)} {algo}")
    Y = evc.synthesize(g["B_rows"], act, layout="frame_major")
    assert_close64(Y, g["Y_rows"], "Y")


@pytest.mark.parametrize("path", [p for p in golden_files("sklearn_") if "l1" not in p and "k50" not in p
                                  and "zero" not in p], ids=os.path.basename)
def test_factorize_surface(path):
    """The drop-in `_factorize(X, W, beta_loss, tol)` / `convert` pair."""
    from exemplars_vc_amd.compat.factorize import _factorize, synthesize_rows
    g = load_golden(path)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H = _factorize(g["X_rows"], g["W_rows"], tol=float(g["tol"]))
    assert H.shape == g["H"].shape
    assert_close64(H, g["H"], "H")
    assert_close64(synthesize_rows(H, g["B_rows"]), g["Y_rows"], "Y")


# ----------------------------------------------------------------------------------------
# the KL loss of `_factorize`'s signature (sklearn beta_loss='kullback-leibler')
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("sklearnkl_"), ids=os.path.basename)
@pytest.mark.parametrize("algo", ["factored", "factored_generic", "fused_c1", "fused_c2"])
def test_sklearn_kl_golden(path, algo):
    import exemplars_vc_amd as evc
    g = load_golden(path)
    tol, max_iter = float(g["tol"]), int(g["max_iter"])
    act, info = evc.solve_activations(
        g["W_rows"], g["X_rows"], layout="frame_major", iters=max_iter, eps_mode="zero_replace",
        init="sklearn", check_every=10 if tol > 0 else 0, stop_rule="sklearn" if tol > 0 else "none",
        tol=tol, info=True, loss="kullback-leibler", **variant_kw(algo))
    assert int(info["n_iter"][0]) == int(g["n_iter"]), (info["n_iter"], g["n_iter"])
    assert_close64(act.T, g["H"], f"KL H {os.path.basename(path)} {algo}")


@pytest.mark.parametrize("N", [1024, 4096])
def test_kl_on_the_register_resident_kernel(N):
    import exemplars_vc_amd as evc
    from exemplars_vc_amd.compat.factorize import _factorize
    o = oracle()
    p = o.synth_problem(25, N, 45, seed=N + 7)
    X = np.ascontiguousarray(p["X"].T)
    X[7] = 0.0
    W = np.ascontiguousarray(p["A"].T)
    want, n, _ = o.sklearn_mu_fixed_dictionary_kl(X, W, 40, 0.0)
    got = evc.solve_activations(W, X, layout="frame_major", iters=40, eps_mode="zero_replace", init="sklearn",
                                loss="kl")
    assert_close64(got, want, "KL resident")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H = _factorize(X, W, beta_loss="kullback-leibler", tol=1e-3, honor_beta_loss=True)
    want, n, _ = o.sklearn_mu_fixed_dictionary_kl(X, W, 150, 1e-3)
    assert_close64(H, want.T, "KL _factorize")
    # unsupported combinations are refused, not silently changed
    with pytest.raises(Exception, match="unsupported"):
        evc.solve_activations(W, X, layout="frame_major", iters=2, eps_mode="add", loss="kl")
    with pytest.raises(Exception, match="unsupported"):
        evc.solve_activations(W, X, layout="frame_major", iters=2, eps_mode="zero_replace", loss="kl", algo="gram")


def test_factorize_batched_utterances_match_single_calls():
    """Per-utterance semantics (own init value, own stop iteration) inside one batch."""
    from exemplars_vc_amd.compat.factorize import factorize_utterances
    o = oracle()
    p = o.synth_problem(25, 96, 0, seed=7)
    rng = np.random.default_rng(3)
    lens = [37, 5, 64, 1, 130]
    Xs = []
    for i, T in enumerate(lens):
        Hs = rng.random((96, T)) * (rng.random((96, T)) < 0.1) * (i + 1)
        Xs.append(np.ascontiguousarray((p["A"] @ Hs + 1e-6).T))
    W = np.ascontiguousarray(p["A"].T)
    tols = 2e-2
    Hs_gpu, n_iter = factorize_utterances(Xs, W, tol=tols)
    for X, Hg, ni in zip(Xs, Hs_gpu, n_iter):
        act, n, _ = o.sklearn_mu_fixed_dictionary(X, W, 150, tols)
        assert n == ni
        assert_close64(Hg, act.T, "batched H")
    assert len(set(int(n) for n in n_iter)) > 1, "test should exercise different stop iterations"
    # hint="latency": the kernels without inter-workgroup exchange (EVC_FLAG_NO_EXCHANGE) - same numbers
    Hs_lat, n_lat = factorize_utterances(Xs, W, tol=tols, hint="latency")
    assert list(n_lat) == list(n_iter)
    for a, b in zip(Hs_lat, Hs_gpu):
        assert_close64(a, b, "latency hint", rtol=1e-10)
    with pytest.raises(ValueError, match="hint"):
        factorize_utterances(Xs, W, tol=tols, hint="fast")


# ----------------------------------------------------------------------------------------
# S2: pymf surface against pymf golden vectors
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("pymf_"), ids=os.path.basename)
@pytest.mark.parametrize("algo", ALGOS)
def test_pymf_surface(path, algo):
    from exemplars_vc_amd.compat.pymf import NMF
    g = load_golden(path)
    mdl = NMF(g["data"].copy(), num_bases=g["W"].shape[1], algo=algo)
    mdl.W = g["W"].copy()
    mdl.H = g["H0"].copy()
    h_id = id(mdl.H)
    mdl.factorize(niter=int(g["niter"]), compute_w=False, compute_err=bool(g["compute_err"]))
    assert id(mdl.H) == h_id, "H must be updated in place"
    assert_close64(mdl.H, g["H"], "H")
    if bool(g["compute_err"]):
        assert len(mdl.ferr) == len(g["ferr"])
        np.testing.assert_allclose(mdl.ferr, g["ferr"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(mdl.frobenius_norm(), float(g["frobenius_norm"]), rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(mdl.residual(), float(g["residual"]), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("path", golden_files("pymfw_"), ids=os.path.basename)
def test_pymf_default_call_updates_the_dictionary_too(path):
    """factorize() with pymf's defaults (compute_w=True, base.py:208): W on the host, H on the GPU, against the
    vendored pymf's own output (tools/make_golden.py)"""
    import warnings
    from exemplars_vc_amd.compat.pymf import NMF
    g = load_golden(path)
    mdl = NMF(g["data"].copy(), num_bases=g["W0"].shape[1])
    mdl.W = g["W0"].copy()
    mdl.H = g["H0"].copy()
    h_id = id(mdl.H)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        mdl.factorize(niter=int(g["niter"]), compute_err=bool(g["compute_err"]))
    assert id(mdl.H) == h_id
    np.testing.assert_allclose(mdl.W, g["W"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(mdl.H, g["H"], rtol=1e-9, atol=1e-300)
    if bool(g["compute_err"]):
        assert len(mdl.ferr) == len(g["ferr"])
        np.testing.assert_allclose(mdl.ferr, g["ferr"], rtol=1e-8)


def test_pymf_doctest_known_answer():
    """pymf/nmf.py:57-63: data=[[1.5],[1.2]], W=I -> H == data."""
    from exemplars_vc_amd.compat.pymf import NMF
    np.random.seed(1234)
    data = np.array([[1.5], [1.2]])
    mdl = NMF(data, num_bases=2)
    mdl.W = np.array([[1.0, 0.0], [0.0, 1.0]])
    mdl.factorize(niter=20, compute_w=False)
    np.testing.assert_allclose(mdl.H, data, rtol=1e-8)
    assert len(mdl.ferr) == 2


# ----------------------------------------------------------------------------------------
# S3: nmf_tool surface (float32, no epsilon) against the restatement - parity unpinned
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS)
def test_nmf_tool_surface(algo):
    from exemplars_vc_amd.compat.nmf_tool import NMF
    o = oracle()
    p = o.synth_problem(40, 48, 33, seed=11)
    rng = np.random.default_rng(5)
    H0 = rng.uniform(0, 1, (48, 33)).astype(np.float32)
    m = NMF(max_iter=60, display_step=10, optimizer="mu", verbose=False, algo=algo)
    W, H = m.fit_transform(p["X"].astype(np.float32), 48, True, p["A"].astype(np.float32), H0=H0)
    want = o.tf_mu_fixed_dictionary(p["X"], p["A"], H0, 60)
    assert H.dtype == np.float32 and H.shape == (48, 33)
    np.testing.assert_allclose(H, want, rtol=2e-3, atol=1e-6 * float(want.max()))
    np.testing.assert_allclose(m.inverse_transform(W, H), W @ H, rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------------------
# every C-ABI mode against the generic oracle statement, ragged shapes
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,T", [(25, 512, 70), (1, 17, 3), (33, 130, 129), (16, 16, 16), (201, 77, 45)])
@pytest.mark.parametrize("eps_mode,eps", [("add", 1e-9), ("zero_replace", 1.1920929e-7), ("none", 0.0),
                                           ("clamp", 1e-15)])
@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
def test_modes_and_layouts(M, N, T, eps_mode, eps, layout):
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, N, T, seed=M * 1000 + N)
    rng = np.random.default_rng(1)
    H0 = rng.random((N, T)) + 1e-4
    code = {"add": o.EPS_ADD, "zero_replace": o.EPS_ZERO_REPLACE, "none": o.EPS_NONE, "clamp": o.EPS_CLAMP}[eps_mode]
    want = o.mu_solve(p["A"], p["X"], H0, 40, eps_mode=code, eps=eps, l1=0.0, algo="gram")
    for algo in VARIANTS:
        kw = variant_kw(algo)
        if layout == "bin_major":
            got = evc.solve_activations(p["A"], p["X"], H0, layout=layout, iters=40, eps_mode=eps_mode,
                                        eps=eps, **kw)
        else:
            got = evc.solve_activations(p["A"].T.copy(), p["X"].T.copy(), H0.T.copy(), layout=layout,
                                        iters=40, eps_mode=eps_mode, eps=eps, **kw).T
        assert_close64(got, want, f"{algo} {eps_mode} {layout}")


@pytest.mark.parametrize("M,Mb,N,T", [(25, 25, 256, 70), (25, 7, 130, 33), (20, 40, 128, 50), (201, 25, 96, 40)])
@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
def test_convert_is_solve_then_synthesize(M, Mb, N, T, layout):
    """evc_nmf_convert (factorize()+convert() in one call) on the fused path (synthesis from the packed
    tiles, Mb <= 32), with a wide target dictionary (Mb > 32) and on the generic path (M > 32)."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, N, T, Mb=Mb, seed=M + Mb)
    H0 = np.random.default_rng(3).random((N, T)) + 1e-4
    Hw = o.mu_solve(p["A"], p["X"], H0, 30, eps_mode=o.EPS_ADD, eps=1e-9)
    Yw = p["B"] @ Hw
    tr = (lambda a: a) if layout == "bin_major" else (lambda a: np.ascontiguousarray(a.T))
    H, Y, info = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), tr(H0), layout=layout, iters=30, info=True)
    assert_close64(tr(H) if layout == "frame_major" else H, Hw, "H")
    assert_close64(tr(Y) if layout == "frame_major" else Y, Yw, "Y")
    assert int(info["n_iter"][0]) == 30
    # Y alone: the activations never leave the solver (constant init, since H0 cannot be passed)
    Hc = evc.solve_activations(tr(p["A"]), tr(p["X"]), layout=layout, iters=12, init="const", init_value=0.02)
    Yc = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), want_h=False, layout=layout, iters=12, init="const",
                     init_value=0.02)
    want = p["B"] @ (Hc if layout == "bin_major" else Hc.T)
    assert_close64(Yc if layout == "bin_major" else Yc.T, want, "Y only")


@pytest.mark.parametrize("N", [1024, 2048, 4096, 1152, 1031, 4001])
@pytest.mark.parametrize("eps_mode,eps,l1", [("add", 1e-9, 0.0), ("zero_replace", 1.1920929e-7, 0.0),
                                              ("clamp", 1e-15, 0.0), ("zero_replace", 1.1920929e-7, 0.3)])
def test_register_resident_kernel_modes(N, eps_mode, eps, l1):
    """k_fused_res runs for N >= 1024 (4, 8 or 16 resident tiles per wavefront; N = 1152 has a streamed
    tail beyond the resident window; N = 1031 / 4001 are padded to a multiple of 128 with zero exemplars,
    which must stay exactly zero and out of the result).  Every guarded mode, ragged T (padded frames), zero
    frames (exact path of the update) and an absorbing zero row, against the oracle."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, T, K = 25, 37, 25
    p = o.synth_problem(M, N, T, seed=N)
    X = p["X"].copy()
    X[:, 5] = 0.0
    H0 = np.random.default_rng(N).random((N, T)) + 1e-4
    H0[17, :] = 0.0
    code = {"add": o.EPS_ADD, "zero_replace": o.EPS_ZERO_REPLACE, "clamp": o.EPS_CLAMP}[eps_mode]
    want = o.mu_solve(p["A"], X, H0, K, eps_mode=code, eps=eps, l1=l1, algo="gram")
    got = evc.solve_activations(p["A"], X, H0, iters=K, eps_mode=eps_mode, eps=eps, l1=l1)
    assert_close64(got, want, f"resident N={N} {eps_mode}")
    assert (got[17] == 0).all() and (got[:, 5] == 0).all()
    gen = evc.solve_activations(p["A"], X, H0, iters=K, eps_mode=eps_mode, eps=eps, l1=l1, fused_c=1)
    assert_close64(gen, want, f"general N={N} {eps_mode}")


@pytest.mark.parametrize("N", [1024, 4096])
def test_register_resident_kernel_with_stop_rules(N):
    """Stopping rules at a size where the resident kernel is in charge: utterances stop at different
    iterations, so some workgroups (16 frames) hold frozen and live frames at once and are handed to
    the general kernel (skip_all_live), including a workgroup straddling two utterances."""
    from exemplars_vc_amd.compat.factorize import factorize_utterances
    o = oracle()
    p = o.synth_problem(25, N, 0, seed=N + 1)
    W = np.ascontiguousarray(p["A"].T)

    def make(kind, T):      # four kinds of utterance that converge at different speeds
        rng = np.random.default_rng(N + kind)
        if kind == 0:
            Hs = rng.random((N, T)) * (rng.random((N, T)) < 3.0 / N)
            return np.ascontiguousarray((p["A"] @ Hs + 1e-6).T)
        if kind == 1:
            return rng.random((T, 25)) ** 4 + 1e-3
        if kind == 2:
            return np.ascontiguousarray(p["A"][:, rng.integers(0, N, T)].T * 3.0)
        return np.ascontiguousarray((p["A"] @ rng.random((N, T))).T)

    Xs = [make(0, 40), make(1, 9), make(2, 70), make(3, 23)]
    tol = 3e-3
    Hs_gpu, n_iter = factorize_utterances(Xs, W, tol=tol, max_iter=60)
    want_iters = []
    for X, Hg, ni in zip(Xs, Hs_gpu, n_iter):
        act, n, _ = o.sklearn_mu_fixed_dictionary(X, W, 60, tol)
        want_iters.append(n)
        assert n == ni, (n, ni)
        assert_close64(Hg, act.T, "H")
    assert len(set(want_iters)) > 2, want_iters


def test_strided_device_tensors_and_zero_iterations():
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 64, 40, seed=2)
    dev = torch.device("cuda")
    Abig = torch.zeros(25, 100, dtype=torch.float64, device=dev)
    Abig[:, :64] = torch.from_numpy(p["A"]).to(dev)
    A = Abig[:, :64]                      # row stride 100
    X = torch.from_numpy(p["X"]).to(dev)
    H0 = torch.rand(64, 40, dtype=torch.float64, device=dev) + 1e-4
    H0c = H0.clone()
    H = evc.solve_activations(A, X, H0, iters=25)
    assert isinstance(H, torch.Tensor) and H.is_cuda
    assert torch.equal(H0, H0c), "caller's H0 must not be clobbered"
    want = o.mu_solve(p["A"], p["X"], H0.cpu().numpy(), 25)
    assert_close64(H.cpu().numpy(), want, "strided")
    H_same = evc.solve_activations(A, X, H0, iters=0)
    assert torch.equal(H_same, H0)


def test_l1_and_zero_frames_and_absorbing_zeros():
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 128, 50, seed=9)
    X = p["X"].copy()
    X[:, :4] = 0.0                          # all-zero frames -> exact zeros, then 0-denominators
    H0 = np.random.default_rng(2).random((128, 50)) + 1e-4
    H0[5, :] = 0.0                          # zeros are absorbing under MU
    for algo in VARIANTS:
        got = evc.solve_activations(p["A"], X, H0, iters=30, eps_mode="zero_replace", l1=0.25,
                                    **variant_kw(algo))
        want = o.mu_solve(p["A"], X, H0, 30, eps_mode=o.EPS_ZERO_REPLACE, eps=o.SK_EPSILON, l1=0.25)
        assert_close64(got, want, "l1")
        assert (got[5] == 0).all() and (got[:, :4] == 0).all()
    # nmf_tool's unguarded form produces NaN exactly where the reference's does
    got = evc.solve_activations(p["A"], X, H0, iters=5, eps_mode="none")
    want = o.mu_solve(p["A"], X, H0, 5, eps_mode=o.EPS_NONE, eps=0.0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(got[:, :4]).any()
    ok = ~np.isnan(want)
    assert_close64(got[ok], want[ok], "none-mode finite part")


def test_frame_residuals_and_error_trace():
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(40, 96, 75, seed=4)
    H0 = np.random.default_rng(8).random((96, 75)) + 1e-4
    H, info = evc.solve_activations(p["A"], p["X"], H0, iters=30, check_every=5, info=True)
    e2 = evc.frame_residuals(p["A"], p["X"], H)
    R = p["X"] - p["A"] @ H
    np.testing.assert_allclose(e2, (R * R).sum(0), rtol=1e-7, atol=1e-20)
    assert info["err"].shape == (1, 7) and np.isnan(info["err"][0, 0])
    Hk = H0.copy()
    for c in range(1, 7):
        Hk = o.mu_solve(p["A"], p["X"], Hk, 5)
        np.testing.assert_allclose(info["err"][0, c], o.residual_fro(p["A"], p["X"], Hk), rtol=1e-7)


# ----------------------------------------------------------------------------------------
# BASELINE configurations at full size: size-independent properties
# ----------------------------------------------------------------------------------------
def test_c2_full_size_properties():
    """C2 (M=25, N=4096, K=100): the oracle needs ~10 s for 688 frames, so only a slice is
    compared entry by entry; the whole batch is checked through properties of MU:
    column independence (a frame's result does not depend on its batch), monotone
    non-increasing residual, non-negativity, and factored == gram algebra."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, T, K = 25, 4096, 2048, 100
    p = o.synth_problem(M, N, T, seed=20190131)
    H, Y = evc.convert(p["A"], p["X"], p["B"], iters=K, eps_mode="zero_replace", init="sklearn",
                       utt_offsets=[0, 688, 1376, T])
    assert (H >= 0).all() and np.isfinite(H).all()
    assert_close64(Y, p["B"] @ H, "Y = B H at full size", rtol=1e-10)
    # frames 0..687 form utterance 0: bitwise identical to solving that utterance alone with the same launch
    # mode (one workgroup per 16 frames); a lone utterance by default takes the cooperative launch, which sums
    # V' in another order
    kw0 = dict(iters=K, eps_mode="zero_replace", init="sklearn")
    H0 = evc.solve_activations(p["A"], p["X"][:, :688], cooperative=False, **kw0)
    Hb = evc.solve_activations(p["A"], p["X"], utt_offsets=[0, 688, 1376, T], cooperative=False, **kw0)
    assert np.array_equal(Hb[:, :688], H0)
    assert_close64(H[:, :688], H0, "cooperative vs one workgroup per tile", rtol=1e-10)
    H0 = evc.solve_activations(p["A"], p["X"][:, :688], **kw0)
    # entry-by-entry against the float64 oracle on 48 frames of utterance 0 (same init value)
    avg = np.sqrt(p["X"][:, :688].mean() / N)
    want = o.mu_solve(p["A"], p["X"][:, :48], np.full((N, 48), avg), K, eps_mode=o.EPS_ZERO_REPLACE,
                      eps=o.SK_EPSILON)
    assert_close64(H[:, :48], want, "C2 slice")
    # gram algebra agrees
    Hg = evc.solve_activations(p["A"], p["X"][:, :688], iters=K, eps_mode="zero_replace", init="sklearn",
                               algo="gram")
    assert_close64(Hg, H0, "gram vs factored")
    # residual decreases monotonically along the iteration
    _, info = evc.solve_activations(p["A"], p["X"][:, :688], iters=K, eps_mode="zero_replace",
                                    init="sklearn", check_every=10, info=True)
    tr = info["err"][0, 1:]
    assert np.all(np.diff(tr) <= 1e-12 * tr[0])


def test_c3_and_c5_shapes_smoke():
    """C3 (M=513, N=8192) and C5 (N=16384 + L1) at reduced T and K: finite, non-negative,
    and the residual decreases; a 16-frame slice is compared with the oracle."""
    import exemplars_vc_amd as evc
    o = oracle()
    for (M, N, T, K, l1) in [(513, 8192, 160, 12, 0.0), (25, 16384, 200, 12, 0.25)]:
        p = o.synth_problem(M, N, T, seed=N)
        H, info = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="const",
                                        init_value=0.01, l1=l1, check_every=4, info=True)
        assert (H >= 0).all() and np.isfinite(H).all()
        tr = info["err"][0, 1:]
        if l1 == 0.0:
            assert np.all(np.diff(tr) <= 0)
        want = o.mu_solve(p["A"], p["X"][:, :16], np.full((N, 16), 0.01), K, eps_mode=o.EPS_ZERO_REPLACE,
                          eps=o.SK_EPSILON, l1=l1)
        assert_close64(H[:, :16], want, f"slice M={M} N={N}")


# ----------------------------------------------------------------------------------------
# factorize(tobe_converted, src_feat) / convert(H, tar_feat, residual): stacking + streams (rows a-9, a-10)
# ----------------------------------------------------------------------------------------
def _feature_files(rng, n_files, bins, frames):
    feats = []
    for T in frames[:n_files]:
        f = {}
        for k, m in bins.items():
            v = rng.random((T, m)) ** 2 + 1e-3
            f[k] = v[:, 0] if m == 1 and k == "f0" else v
        feats.append(f)
    return feats


@pytest.mark.parametrize("use_stft", [True, False])
def test_factorize_and_convert_feature_level(use_stft):
    import warnings
    from exemplars_vc_amd.compat.factorize import factorize, convert
    o = oracle()
    rng = np.random.default_rng(12)
    if use_stft:
        bins = {"real": 201}
        src = _feature_files(rng, 3, bins, [30, 41, 25])
        tar = _feature_files(rng, 3, bins, [30, 41, 25])
        for f in src + tar:
            f["real"] = f["real"] * np.sign(rng.standard_normal(f["real"].shape))     # |.| is applied by the path
        conv = {"real": _feature_files(rng, 1, bins, [37])[0]["real"]}
    else:
        bins = {"sp": 65, "ap": 65, "f0": 1}
        src = _feature_files(rng, 3, bins, [30, 41, 25])
        tar = _feature_files(rng, 3, bins, [30, 41, 25])
        conv = _feature_files(rng, 1, bins, [37])[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H, R = factorize(conv, src, use_stft=use_stft, tol=1e-3)
        out = convert(H, tar, R, use_stft=use_stft)
        Hw, Rw = o.s_factorize_features(conv, src, use_stft, tol=1e-3)
        outw = o.s_convert_features(Hw, tar, Rw, use_stft)
    for k in Hw:
        assert H[k].shape == Hw[k].shape
        assert_close64(H[k], Hw[k], k)
    if use_stft:
        assert R is None and out.shape == (37, 201)
        assert_close64(out, outw, "converted stft")
    else:
        for k in Rw:
            # the reference's residual log(H.T A - conv) is NaN wherever the reconstruction undershoots; where
            # the fit is exact to rounding (the one-bin f0 stream) that sign is rounding noise, so the
            # comparison is made on exp(r) = H.T A - conv, ignoring differences below 1e-10 of the data scale
            scale = float(np.max(np.abs(np.asarray(conv[k[2:]]))))
            dg = np.where(np.isnan(R[k]), 0.0, np.exp(R[k]))
            dw = np.where(np.isnan(Rw[k]), 0.0, np.exp(Rw[k]))
            np.testing.assert_allclose(dg, dw, rtol=1e-6, atol=1e-10 * scale)
        assert set(out) == {"sp", "ap", "f0"} and out["f0"].shape == (37,)


def test_degenerate_sizes_and_empty_utterances():
    import exemplars_vc_amd as evc
    o = oracle()
    # one bin, one exemplar, one frame
    A, X, H0 = np.array([[2.0]]), np.array([[3.0]]), np.array([[0.7]])
    for algo in VARIANTS:
        got = evc.solve_activations(A, X, H0, iters=9, **variant_kw(algo))
        assert_close64(got, o.mu_solve(A, X, H0, 9), algo)
    # a batch with empty utterances in it (first, middle, last)
    p = o.synth_problem(25, 64, 50, seed=77)
    X_rows, W_rows = np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T)
    offs = [0, 0, 20, 20, 50, 50]
    act, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=30, eps_mode="zero_replace",
                                      init="sklearn", check_every=10, stop_rule="sklearn", tol=1e-3,
                                      utt_offsets=offs, info=True)
    for (a, b), ni in zip([(0, 20), (20, 50)], [info["n_iter"][1], info["n_iter"][3]]):
        want, n, _ = o.sklearn_mu_fixed_dictionary(X_rows[a:b], W_rows, 30, 1e-3)
        assert n == ni
        assert_close64(act[a:b], want, "utterance")
    assert np.all(np.isnan(info["err"][[0, 2, 4]]))
    # no frames at all
    e = evc.solve_activations(W_rows, X_rows[:0], layout="frame_major", iters=5, init="sklearn")
    assert e.shape == (0, 64)


def test_float32_factorize_surface():
    """float32 inputs keep their dtype through `_factorize`, as they do through scikit-learn (check_array
    keeps float32); compared with the float32 oracle run.  Tolerance: float32 trajectories of different
    summation orders drift apart at ~K*sqrt(N)*6e-8, hence rtol 5e-3 with an absolute floor of 1e-5*max|H|."""
    import warnings
    from exemplars_vc_amd.compat.factorize import _factorize
    o = oracle()
    p = o.synth_problem(40, 96, 60, seed=21)
    X = np.ascontiguousarray(p["X"].T).astype(np.float32)
    W = np.ascontiguousarray(p["A"].T).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H = _factorize(X, W, tol=0.0)
    want, n, _ = o.sklearn_mu_fixed_dictionary(X, W, 150, 0.0)
    assert H.dtype == np.float32 and want.dtype == np.float32
    np.testing.assert_allclose(H, want.T, rtol=5e-3, atol=1e-5 * float(want.max()))


@pytest.mark.parametrize("N,T", [(4096, 688), (4096, 37), (2048, 100), (16384, 20), (4001, 50)])
@pytest.mark.parametrize("eps_mode,eps", [("zero_replace", 1.1920929e-7), ("add", 1e-9)])
def test_cooperative_launch_matches_single_workgroup_tiles(N, T, eps_mode, eps):
    """One or two utterances leave most CUs idle with one workgroup per 16 frames; k_fused_res then runs
    cooperatively (several workgroups share a frame tile, split the exemplars and exchange V' through
    global memory every iteration).  Same result as the non-cooperative launch up to summation order, and
    both against the oracle; zero frames and an absorbing zero row included."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, K = 25, 30
    p = o.synth_problem(M, N, T, seed=N + T)
    X = p["X"].copy()
    X[:, 3] = 0.0
    H0 = np.random.default_rng(N).random((N, T)) + 1e-4
    H0[5, :] = 0.0
    kw = dict(iters=K, eps_mode=eps_mode, eps=eps)
    coop = evc.solve_activations(p["A"], X, H0, **kw)
    solo = evc.solve_activations(p["A"], X, H0, cooperative=False, **kw)
    assert_close64(coop, solo, f"coop vs solo N={N} T={T}", rtol=1e-11)
    assert (coop[5] == 0).all() and (coop[:, 3] == 0).all()
    if N * T <= 4096 * 100:          # the oracle's Gram product is the slow part
        code = {"add": o.EPS_ADD, "zero_replace": o.EPS_ZERO_REPLACE}[eps_mode]
        want = o.mu_solve(p["A"], X, H0, K, eps_mode=code, eps=eps, algo="factored")
        assert_close64(coop, want, f"coop vs oracle N={N} T={T}")


def test_cooperative_launch_with_stop_rule_and_two_utterances():
    """The default call (_factorize: check every 10 iterations, tol) on the cooperative path: launches of 10
    iterations, per-utterance freezing between them."""
    import warnings
    from exemplars_vc_amd.compat.factorize import factorize_utterances
    o = oracle()
    M, N = 25, 4096
    rows = [np.ascontiguousarray(o.synth_problem(M, N, T, seed=T)["X"].T) for T in (70, 45)]
    W = np.ascontiguousarray(o.synth_problem(M, N, 8, seed=1)["A"].T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Hs, n_iter = factorize_utterances(rows, W, tol=2e-3)
    for X_rows, H, n in zip(rows, Hs, n_iter):
        act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X_rows, W, 150, 2e-3)
        assert int(n) == n_ref
        assert_close64(H, act.T, "coop + stop rule")


@pytest.mark.parametrize("Mb,N,T", [(25, 4096, 688), (1, 17, 3), (16, 100, 16), (33, 130, 50), (64, 257, 31),
                                    (65, 96, 40), (201, 300, 45)])
@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_synthesize_alone(Mb, N, T, layout, dt):
    """evc_synthesize (np.matmul(H.T, B), 04_align_n_nmf.py:391) on caller memory: the skinny kernel
    (Mb <= 64: wavefronts split the exemplars) and the strided MFMA fallback, both layouts and dtypes,
    ragged sizes, plus a strided (sliced) activation matrix."""
    import torch
    import exemplars_vc_amd as evc
    rng = np.random.default_rng(Mb * 1000 + N + T)
    B = rng.random((Mb, N)).astype(dt)
    Hfull = rng.random((N, T + 5)).astype(dt)
    H = Hfull[:, :T]
    want = B.astype(np.float64) @ H.astype(np.float64)
    tol = 1e-12 if dt == np.float64 else 2e-5
    if layout == "bin_major":
        Y = evc.synthesize(B, np.ascontiguousarray(H), layout=layout)
        assert Y.shape == (Mb, T) and Y.dtype == dt
        assert np.abs(Y - want).max() <= tol * np.abs(want).max()
        # a strided view on the device: columns 0..T-1 of a wider matrix (row stride T + 5)
        Hd = torch.as_tensor(Hfull).cuda()[:, :T]
        Yd = evc.synthesize(torch.as_tensor(B).cuda(), Hd, layout=layout).cpu().numpy()
        assert np.abs(Yd - want).max() <= tol * np.abs(want).max()
    else:
        Y = evc.synthesize(np.ascontiguousarray(B.T), np.ascontiguousarray(H.T), layout=layout)
        assert Y.shape == (T, Mb) and Y.dtype == dt
        assert np.abs(Y.T - want).max() <= tol * np.abs(want).max()


@pytest.mark.parametrize("N", [130, 1024])
@pytest.mark.parametrize("eps_mode,eps", [("add", 1e-9), ("none", 0.0), ("zero_replace", 1.1920929e-7)])
@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
def test_float32_small_bin_count_rides_the_float64_kernels(N, eps_mode, eps, layout):
    """float32 callers with M <= 32 are widened, solved by the float64 fused kernels and narrowed on the
    way out: the result is the float64 solution of the float32 inputs, rounded once (6e-8), for H and Y,
    in both layouts, with a given start; the float32 generic path (fused=False) stays available and agrees
    to float32 accuracy."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, Mb, T, K = 25, 20, 70, 40
    p = o.synth_problem(M, N, T, Mb=Mb, seed=N)
    A, X, B = (p[k].astype(np.float32) for k in ("A", "X", "B"))
    H0 = (np.random.default_rng(N).random((N, T)) + 1e-4).astype(np.float32)
    code = {"add": o.EPS_ADD, "none": o.EPS_NONE, "zero_replace": o.EPS_ZERO_REPLACE}[eps_mode]
    want = o.mu_solve(A.astype(np.float64), X.astype(np.float64), H0.astype(np.float64), K, eps_mode=code, eps=eps)
    want_y = B.astype(np.float64) @ want
    tr = (lambda a: a) if layout == "bin_major" else (lambda a: np.ascontiguousarray(a.T))
    H, Y = evc.convert(tr(A), tr(X), tr(B), tr(H0), layout=layout, iters=K, eps_mode=eps_mode, eps=eps)
    assert H.dtype == np.float32 and Y.dtype == np.float32
    H, Y = (H, Y) if layout == "bin_major" else (H.T, Y.T)
    np.testing.assert_allclose(H, want, rtol=2e-7, atol=1e-30)
    np.testing.assert_allclose(Y, want_y, rtol=2e-7)
    Hg = evc.solve_activations(tr(A), tr(X), tr(H0), layout=layout, iters=K, eps_mode=eps_mode, eps=eps, fused=False)
    Hg = Hg if layout == "bin_major" else Hg.T
    assert Hg.dtype == np.float32
    np.testing.assert_allclose(Hg, want, rtol=5e-3, atol=1e-5 * float(want.max()))
    # sklearn initialisation and the per-utterance stop rule travel through the widened call as well
    Hs, info = evc.solve_activations(tr(A), tr(X), layout=layout, iters=60, eps_mode="zero_replace", init="sklearn",
                                     check_every=10, stop_rule="sklearn", tol=1e-3, info=True)
    act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X.T.astype(np.float64), A.T.astype(np.float64), 60, 1e-3)
    assert int(info["n_iter"][0]) == n_ref
    np.testing.assert_allclose(Hs if layout == "bin_major" else Hs.T, act.T, rtol=2e-7, atol=1e-30)


def test_cooperative_timeout_falls_back_to_one_workgroup_per_tile():
    """A cooperative launch whose wait budget ran out (another process holding the CUs its peers needed)
    voids its results; the solve is then redone non-cooperatively from the untouched inputs.  The flag is
    raised artificially here: same H, n_iter and error trace as a call that never was cooperative."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 4096, 90, seed=5)
    H0 = np.random.default_rng(5).random((4096, 90)) + 1e-4
    for kw in (dict(H0=H0, iters=30, eps_mode="add"),
               dict(H0=None, iters=40, eps_mode="zero_replace", init="sklearn", check_every=10, stop_rule="sklearn",
                    tol=1e-3, utt_offsets=[0, 50, 90])):
        h0 = kw.pop("H0")
        want, iw = evc.solve_activations(p["A"], p["X"], h0, cooperative=False, info=True, **kw)
        # True: the flag is up from the start; 2: it goes up in front of the third launch of the iteration loop (the
        # second configuration runs one launch per 10 iterations), i.e. the redo follows a partly completed solve
        for when in ((True, 2) if kw.get("check_every") else (True,)):
            got, ig = evc.solve_activations(p["A"], p["X"], h0, _fake_coop_timeout=when, info=True, **kw)
            assert ig["redo"] == 1 and ig["exchange"] == 0, ig      # the library says that it redid the solve
            assert iw["redo"] == 0
            assert np.array_equal(got, want), when
            assert np.array_equal(ig["n_iter"], iw["n_iter"]), when
            assert np.array_equal(np.nan_to_num(ig["err"]), np.nan_to_num(iw["err"])), when


def test_two_processes_sharing_the_gpu_with_cooperative_launches(tmp_path):
    """Two processes on one GPU, each issuing a few solves whose workgroups exchange through device memory: both
    must finish with correct results (every wait is bounded, and a timed-out solve is redone without exchange -
    that path itself is tested deterministically by test_cooperative_timeout_falls_back_...).  Three solves each:
    an integration check of two contexts on the card, not a stress loop."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        "import exemplars_vc_amd as evc\n"
        "from oracle import evc_oracle as o\n"
        "p = o.synth_problem(25, 4096, 688, seed=int(sys.argv[1]))\n"
        "want = evc.solve_activations(p['A'], p['X'], iters=20, eps_mode='zero_replace', init='sklearn', cooperative=False)\n"
        "worst = 0.0\n"
        "for i in range(3):\n"
        "    got = evc.solve_activations(p['A'], p['X'], iters=20, eps_mode='zero_replace', init='sklearn')\n"
        "    worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))\n"
        "print('WORST', worst)\n"
        "assert worst < 1e-10, worst\n")
    procs = [subprocess.Popen([sys.executable, str(script), str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for k in (1, 2)]
    outs = [pr.communicate(timeout=600)[0].decode() for pr in procs]
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0 and "WORST" in out, out[-2000:]


@pytest.mark.parametrize("scale", [1e-200, 1e-140, 1e140, 1e200])
@pytest.mark.parametrize("N,T", [(130, 40), (1024, 40), (4096, 300)])
def test_extreme_magnitudes_take_the_exact_path(scale, N, T):
    """Denominators outside [2^-250, 2^250] leave the fast path of the fused update (batch inversion would
    overflow or underflow there): the exact path must give the reference's result at any magnitude float64
    can hold.  General kernel (N=130), register-resident kernel (N=1024, T=300 -> one workgroup per tile)
    and its cooperative launch (N=4096, T=300 is 19 tiles)."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, N, T, seed=N)
    X = p["X"] * scale
    K = 12
    want, _, _ = o.sklearn_mu_fixed_dictionary(np.ascontiguousarray(X.T), np.ascontiguousarray(p["A"].T), K, 0.0)
    got = evc.solve_activations(p["A"], X, iters=K, eps_mode="zero_replace", init="sklearn")
    assert np.isfinite(got).all()
    assert_close64(got, want.T, f"scale {scale:g} N={N}")


@pytest.mark.parametrize("N", [130, 1024, 4096])
def test_nan_and_inf_frames_propagate_like_the_reference(N):
    """pymf and nmf_tool validate nothing: a NaN or infinite frame yields NaN activations for that frame and
    leaves every other frame untouched (frames are independent columns).  Same NaN pattern, same finite
    values, on the general, register-resident and cooperative kernels."""
    import exemplars_vc_amd as evc
    o = oracle()
    T, K = 50, 8
    p = o.synth_problem(25, N, T, seed=3 * N)
    X = p["X"].copy()
    X[:, 7] = np.nan
    X[3, 21] = np.inf
    H0 = np.random.default_rng(N).random((N, T)) + 1e-4
    with np.errstate(all="ignore"):
        want = o.mu_solve(p["A"], X, H0, K, eps_mode=o.EPS_ADD, eps=1e-9)
    got = evc.solve_activations(p["A"], X, H0, iters=K, eps_mode="add", eps=1e-9)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(got[:, 7]).all() and np.isnan(got[:, 21]).all()      # inf * 0-weight bins -> NaN after one step
    ok = ~np.isnan(want)
    r = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
    assert r.max() <= RTOL64


def test_more_than_2_to_31_activations():
    """N x T beyond 2^31 elements (17 GB of float64 activations, all on the device): every index in the pack,
    solve, export and synthesis kernels must be 64-bit.  Frames are independent columns, so the first and the
    last 48 frames of the big batch must equal the same frames solved alone (same launch mode, bitwise)."""
    import torch
    import exemplars_vc_amd as evc
    N, M, T, K = 16384, 25, 131200, 3
    assert N * T > 2 ** 31
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free device memory")
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(5)
    A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    B = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64)
    X = torch.rand(T, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
    kw = dict(layout="frame_major", iters=K, eps_mode="zero_replace", init="const", init_value=0.01,
              cooperative=False)
    H, Y = evc.convert(A, X, B, **kw)
    assert H.shape == (T, N) and Y.shape == (T, M)
    for sl in (slice(0, 48), slice(T - 48, T), slice(65536 + 16, 65536 + 64)):
        Hs, Ys = evc.convert(A, X[sl].contiguous(), B, **kw)
        assert torch.equal(H[sl], Hs)
        assert torch.allclose(Y[sl], Ys, rtol=1e-12, atol=0)
    assert bool(torch.isfinite(H[-1]).all()) and float(H.min()) >= 0.0
    del H, Y
    evc.release_workspaces()
    torch.cuda.empty_cache()
