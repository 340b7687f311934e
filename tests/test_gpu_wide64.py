"""k_fused_wide64 (evc_wide64.hip): the fused FACTORED update for wide float64 spectra (144 < M <= 528 bins; the
513-bin STFT magnitudes of BASELINE C3 / C5_513, and from round 4 the 201-bin |Re STFT| flow in float64).

The library routes batches of 240 ... 1000 frame tiles to it (6 to ~20 utterances: DESIGN.md section 5.2b, where it
beats the two-contraction path); elsewhere the tuning bits select it: `fused_w >= 3` (the narrowest instance of 3, 4, 5,
7 or 8 whole bin tiles per wavefront, plus one tile split over the four, that holds M) and / or `fused_c` (exemplar ranges per frame group).  Every case enters through the C ABI
and is compared with the float64 oracle on the same inputs; tolerance 1e-9 relative (summation order only: the
quotient is the correctly rounded division)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def oracle():
    from oracle import evc_oracle as o
    return o


def check(got, want, rtol=RTOL):
    want = np.asarray(want, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=rtol, atol=1e-12 * float(np.abs(want).max()))


def sk_want(o, A, X, K, l1=0.0):
    N = A.shape[1]
    return o.mu_solve(A, X, np.full((N, X.shape[1]), np.sqrt(X.mean() / N)), K, eps_mode=o.EPS_ZERO_REPLACE,
                      eps=float(np.finfo(np.float64).eps), l1=l1, algo="factored")


@pytest.mark.parametrize("M,N,T,K,c,tpw", [
    (513, 256, 64, 12, 0, 8), (513, 256, 64, 12, 1, 8), (513, 256, 40, 12, 2, 8), (513, 250, 50, 12, 3, 8),
    (513, 1000, 100, 8, 0, 8), (513, 1000, 100, 8, 6, 8), (257, 300, 70, 10, 0, 4), (400, 512, 33, 10, 5, 7),
    (528, 200, 17, 10, 0, 8), (209, 128, 32, 10, 2, 4), (320, 512, 130, 8, 8, 5), (448, 130, 1, 8, 0, 7),
    (513, 17, 5, 6, 0, 8), (272, 256, 48, 8, 2, 4), (273, 256, 48, 8, 2, 5), (512, 256, 48, 8, 3, 8),
    (201, 256, 64, 12, 0, 3), (201, 300, 50, 10, 2, 3), (145, 128, 40, 8, 1, 3), (208, 200, 33, 10, 3, 3), (201, 1000, 100, 8, 6, 3),
    (193, 64, 1, 6, 0, 3),
])
def test_wide64_kernel_against_the_oracle(M, N, T, K, c, tpw):
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, N, T, seed=M + N + T)
    got, info = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", fused_c=c,
                                      fused_w=min(tpw, 4), info=True)
    assert info["kernel"] == "k_fused_wide64" and info["launches"] == 1 and info["redo"] == 0, info
    if c:
        assert info["members"] == min(c, max(1, ((N + 15) // 16) // 2)), info
    assert got.dtype == np.float64
    check(got, sk_want(o, p["A"], p["X"], K))


def test_wide64_routing_by_batch_size():
    """(evc_api.hip, use_wide: the fused kernel serves 240 ... 1000 frame tiles, where it beats the two contractions)"""
    import exemplars_vc_amd as evc
    o = oracle()
    for T, kernel in ((64, "k_gemm_nt"), (4000, "k_fused_wide64")):
        p = o.synth_problem(513, 256, T, seed=T)
        got, info = evc.solve_activations(p["A"], p["X"], iters=4, eps_mode="zero_replace", init="sklearn", info=True)
        assert info["kernel"] == kernel, info
        check(got, sk_want(o, p["A"], p["X"], 4))
    # 3 whole bin tiles per wavefront (176 < M <= 208): from 100 frame tiles on at N >= 2048, from 43 below
    for M, T, kernel in ((201, 4000, "k_fused_wide64"), (201, 500, "k_gemm_nt"), (170, 4000, "k_gemm_nt")):
        p = o.synth_problem(M, 256, T, seed=T + M)
        got, info = evc.solve_activations(p["A"], p["X"], iters=4, eps_mode="zero_replace", init="sklearn", info=True)
        assert info["kernel"] == kernel, info
        check(got, sk_want(o, p["A"], p["X"], 4))


@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
@pytest.mark.parametrize("eps_mode,eps", [("add", 1e-9), ("none", 0.0), ("clamp", 1e-15)])
def test_wide64_other_surfaces_and_convert(layout, eps_mode, eps):
    """pymf / nmf_tool / deComP semantics with a given H0, the synthesis Y = B H, both orientations"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(513, 512, 90, seed=4)
    A, X, B = p["A"], p["X"], p["B"]
    H0 = np.random.default_rng(0).random((512, 90)) + 1e-4
    mode = {"add": o.EPS_ADD, "none": o.EPS_NONE, "clamp": o.EPS_CLAMP}[eps_mode]
    want = o.mu_solve(A, X, H0, 15, eps_mode=mode, eps=eps, algo="factored")
    tr = (lambda z: z) if layout == "bin_major" else (lambda z: np.ascontiguousarray(z.T))
    H, Y = evc.convert(tr(A), tr(X), tr(B), tr(H0), layout=layout, iters=15, eps_mode=eps_mode, eps=eps, fused_w=4)
    H, Y = (H, Y) if layout == "bin_major" else (H.T, Y.T)
    check(H, want)
    check(Y, B @ want)


def test_wide64_stop_rule_l1_and_utterances():
    """scikit-learn's stop rule evaluated between launches on the published V; an utterance that stops early is
    frozen while its neighbour in the same frame group goes on (37 frames: the boundary cuts a group of 32); l1"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(513, 384, 90, seed=9)
    A, X = p["A"], p["X"].copy()
    rng = np.random.default_rng(3)
    X[:, :37] = A[:, rng.integers(0, 384, 37)] * rng.random(37) + 1e-9        # near-exemplar frames: converge early
    X_rows, W_rows = np.ascontiguousarray(X.T), np.ascontiguousarray(A.T)
    offs = np.array([0, 37, 90], dtype=np.int32)
    H, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=120, eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=5e-3, info=True,
                                    utt_offsets=offs, fused_w=4)
    assert info["kernel"] == "k_fused_wide64"
    n_its = []
    for u in range(2):
        a, b = offs[u], offs[u + 1]
        act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X_rows[a:b], W_rows, 120, 5e-3)
        assert int(info["n_iter"][u]) == n_ref
        n_its.append(n_ref)
        check(H[a:b], act, rtol=1e-8)
    assert n_its[0] < n_its[1], "the first utterance was meant to stop before the second"
    got = evc.solve_activations(p["A"], p["X"], iters=10, eps_mode="zero_replace", init="sklearn", l1=0.05, fused_w=4)
    check(got, sk_want(o, p["A"], p["X"], 10, l1=0.05))


def test_wide64_repeatable_and_prepared_dictionary():
    """bitwise the same from call to call (partials are summed in range order), and from a prepared dictionary"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(513, 640, 150, seed=12)
    kw = dict(iters=8, eps_mode="zero_replace", init="sklearn", fused_c=3, fused_w=4)
    a = evc.solve_activations(p["A"], p["X"], **kw)
    b = evc.solve_activations(p["A"], p["X"], **kw)
    assert np.array_equal(a, b)
    pd = evc.prepare_dictionary(p["A"], p["B"])
    c, info = evc.solve_activations(pd, p["X"], info=True, **kw)
    assert info["kernel"] == "k_fused_wide64" and info["prepared"] == 1, info
    assert np.array_equal(a, c)
    # the first 64 frames alone are the first 64 columns of the batch (frame groups are independent)
    d = evc.solve_activations(p["A"], np.ascontiguousarray(p["X"][:, :64]), iters=8, eps_mode="zero_replace",
                              init_value=float(np.sqrt(p["X"].mean() / 640)), init="const", fused_c=3, fused_w=4)
    e = evc.solve_activations(p["A"], p["X"], iters=8, eps_mode="zero_replace",
                              init_value=float(np.sqrt(p["X"].mean() / 640)), init="const", fused_c=3, fused_w=4)
    assert np.array_equal(d, e[:, :64])


@pytest.mark.parametrize("dtype,M,kernel", [(np.float64, 513, "k_gemm_nt"), (np.float32, 201, "k_gemm2")])
def test_generic_path_gate_after_every_utterance_stopped(dtype, M, kernel):
    """the two-contraction path: once the stop rule has stopped every utterance the launches still queued return at
    once (evc_api.hip, `gate`); an utterance that goes on keeps the gate open for the whole batch"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, 256, 90, seed=21)
    A, X = p["A"].astype(dtype), p["X"].astype(dtype).copy()
    rng = np.random.default_rng(3)
    X[:, :37] = (A[:, rng.integers(0, 256, 37)] * rng.random(37).astype(dtype) + dtype(1e-9)).astype(dtype)
    X_rows, W_rows = np.ascontiguousarray(X.T), np.ascontiguousarray(A.T)
    X64, W64 = X_rows.astype(np.float64), W_rows.astype(np.float64)
    eps32 = dict(rtol=2e-3, atol=1e-6) if dtype == np.float32 else dict(rtol=1e-8, atol=1e-300)
    for offs in (np.array([0, 37], dtype=np.int32), np.array([0, 37, 90], dtype=np.int32)):
        T = int(offs[-1])
        H, info = evc.solve_activations(W_rows, X_rows[:T], layout="frame_major", iters=120, eps_mode="zero_replace",
                                        init="sklearn", check_every=10, stop_rule="sklearn", tol=5e-3, info=True,
                                        utt_offsets=offs)
        assert info["kernel"] == kernel, info
        for u in range(len(offs) - 1):
            a, b = offs[u], offs[u + 1]
            act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X64[a:b], W64, 120, 5e-3)
            n_got = int(info["n_iter"][u])
            if dtype == np.float64:
                assert n_got == n_ref
            else:
                # a float32 residual can cross the threshold one check earlier or later than the float64 oracle's:
                # within one check, and the activations are compared with the oracle run for exactly n_got iterations
                assert abs(n_got - n_ref) <= 10, (n_got, n_ref)
                if n_got != n_ref:
                    act, _, _ = o.sklearn_mu_fixed_dictionary(X64[a:b], W64, n_got, 0.0)
            got = H[a:b].astype(np.float64)
            np.testing.assert_allclose(got, act, atol=eps32["atol"] * float(np.abs(act).max()) if dtype == np.float32 else 0.0,
                                       rtol=eps32["rtol"])


@pytest.mark.parametrize("seed", range(14))
def test_wide64_random_shapes_against_the_two_contraction_path(seed):
    """Differential test over seeded random shapes: the fused float64 kernel (random ranges per group, every instance)
    against the two-contraction path on the same call - ragged utterances, exemplar counts that are not multiples of
    16, bin counts on both sides of the whole-tile boundaries, both layouts, the synthesis, error traces over several
    launches, correctly rounded quotients on request."""
    import exemplars_vc_amd as evc
    o = oracle()
    rng = np.random.default_rng(700 + seed)
    M = int(rng.choice([int(rng.integers(209, 529)), 64 * int(rng.integers(4, 9)) + int(rng.integers(0, 17)), 513, 257]))
    M = min(max(M, 209), 528)
    N = int(rng.choice([int(rng.integers(40, 300)), int(rng.integers(300, 1200)), 16 * int(rng.integers(8, 60))]))
    lens = [int(rng.integers(1, 150)) for _ in range(int(rng.integers(1, 6)))]
    T = sum(lens)
    p = o.synth_problem(M, N, T, seed=1300 + seed)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    K = int(rng.integers(1, 14))
    layout = "frame_major" if seed % 2 else "bin_major"
    tr = (lambda a: np.ascontiguousarray(a.T)) if layout == "frame_major" else (lambda a: np.ascontiguousarray(a))
    kw = dict(layout=layout, iters=K, eps_mode=["zero_replace", "add", "clamp", "none"][seed % 4],
              init=["sklearn", "const"][seed % 2], utt_offsets=offs, exact_div=bool(seed % 3 == 0))
    if kw["init"] == "const":
        kw["init_value"] = 0.21
    if seed % 4 == 3:
        kw.update(check_every=4, info=True)
    if seed % 5 == 1:
        kw.update(l1=0.03)
    c = int(rng.integers(0, 9))
    got = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused_c=c, fused_w=4, **kw)
    want = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused=False, **kw)
    for g, wv, name in zip(got[:2], want[:2], ("H", "Y")):
        np.testing.assert_allclose(g, wv, rtol=1e-9, atol=1e-12 * float(np.abs(wv).max()),
                                   err_msg=f"seed {seed} M={M} N={N} lens={lens} K={K} c={c} {layout}: {name}")
    if seed % 4 == 3:
        assert got[2]["kernel"] == "k_fused_wide64" and want[2]["kernel"] == "k_gemm_nt"
        np.testing.assert_allclose(got[2]["err"], want[2]["err"], rtol=1e-9, atol=1e-12 * float(np.linalg.norm(p["X"])),
                                   equal_nan=True)
