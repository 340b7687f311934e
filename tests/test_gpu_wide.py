"""k_fused_wide (evc_wide.hip): the fused FACTORED update for wide float32 spectra (32 < M <= 208 bins) - the
|Re STFT| stream the reference script runs by default (04_align_n_nmf.py:315-326, config/config:12).

Every case enters through the C ABI (evc_nmf_solve / evc_nmf_convert) and is compared with the float64 oracle run
on the same float32 inputs.  Tolerance: float32 trajectories of different summation orders drift apart at about
K sqrt(N) 6e-8, hence rtol 2e-3 with an absolute floor of 1e-6 max|H| (north_star asks 1e-4 of the float64 path).
The exemplar ranges per frame group (c) and the wavefronts per workgroup (W) are forced through the tuning bits so
that every dependency pattern of the task queue runs: no split, the direct sum of 2-4 partials, the reduce tasks.
(Small batches are routed to the two-contraction path by default - evc_api.hip, use_wide - so the cases here force
the fused kernel through those bits; `fused_w=4` alone means "this kernel, automatic ranges".)"""
import numpy as np
import pytest

from conftest import GOLDEN, load_golden  # noqa: F401

pytestmark = pytest.mark.gpu
RTOL, AFLOOR = 2e-3, 1e-6


def oracle():
    from oracle import evc_oracle as o
    return o


def check(got, want, rtol=RTOL):
    want = np.asarray(want, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=rtol, atol=AFLOOR * float(np.abs(want).max()))


def sk_want(o, A32, X32, K, l1=0.0):
    """the scikit-learn update in float64 on the float32 inputs (bins as rows)"""
    N = A32.shape[1]
    X64 = X32.astype(np.float64)
    return o.mu_solve(A32.astype(np.float64), X64, np.full((N, X32.shape[1]), np.sqrt(X64.mean() / N)), K,
                      eps_mode=o.EPS_ZERO_REPLACE, eps=float(np.finfo(np.float32).eps), l1=l1, algo="factored")


@pytest.mark.parametrize("M,N,T,K,c,w", [
    (201, 256, 64, 20, 0, 4), (201, 256, 64, 20, 1, 4), (201, 256, 64, 20, 2, 8), (201, 250, 50, 20, 3, 4),
    (201, 1000, 688, 30, 0, 4), (201, 1000, 688, 30, 6, 4), (201, 1000, 100, 30, 8, 8), (201, 1000, 100, 30, 4, 8),
    (33, 300, 100, 25, 0, 4), (64, 512, 130, 25, 4, 0), (100, 512, 130, 25, 5, 0), (150, 200, 33, 25, 0, 8),
    (208, 4096, 688, 20, 0, 8), (201, 17, 5, 10, 0, 4), (201, 4096, 1, 15, 0, 8), (201, 512, 5000, 12, 0, 0),
])
def test_wide_kernel_against_the_oracle(M, N, T, K, c, w):
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, N, T, seed=M + N + T)
    A32, X32 = p["A"].astype(np.float32), p["X"].astype(np.float32)
    got, info = evc.solve_activations(A32, X32, iters=K, eps_mode="zero_replace", init="sklearn", fused_c=c, fused_w=w,
                                      info=True)
    assert info["kernel"] == "k_fused_wide" and info["launches"] == 1 and info["redo"] == 0, info
    if c:
        assert info["members"] == min(c, max(1, ((N + 15) // 16) // 2)), info
    assert got.dtype == np.float32
    check(got, sk_want(o, A32, X32, K))


@pytest.mark.parametrize("layout", ["bin_major", "frame_major"])
@pytest.mark.parametrize("eps_mode,eps", [("add", 1e-9), ("none", 0.0), ("clamp", 1e-15)])
def test_wide_other_surfaces_and_convert(layout, eps_mode, eps):
    """pymf / nmf_tool / deComP guards with a given start, both orientations, H and Y = B H from one call"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 500, 90, seed=4)
    A32, X32, B32 = (p[k].astype(np.float32) for k in ("A", "X", "B"))
    H0 = (np.random.default_rng(0).random((500, 90)) + 1e-4).astype(np.float32)
    want = o.mu_solve(A32.astype(np.float64), X32.astype(np.float64), H0.astype(np.float64), 30,
                      eps_mode={"add": o.EPS_ADD, "none": o.EPS_NONE, "clamp": o.EPS_CLAMP}[eps_mode], eps=eps,
                      algo="factored")
    tr = (lambda z: z) if layout == "bin_major" else (lambda z: np.ascontiguousarray(z.T))
    H, Y, info = evc.convert(tr(A32), tr(X32), tr(B32), tr(H0), layout=layout, iters=30, eps_mode=eps_mode, eps=eps,
                             info=True, fused_w=4)
    assert info["kernel"] == "k_fused_wide"
    H, Y = (H, Y) if layout == "bin_major" else (H.T, Y.T)
    check(H, want)
    check(Y, B32.astype(np.float64) @ want)


def test_wide_l1_and_the_stop_rule_per_utterance():
    """the scikit-learn call of 04_align_n_nmf.py:212 on float32 rows: tol = 1e-3 stop test every 10 iterations, three
    utterances in one batch, each with its own start value, error trace and n_iter; then the L1 variant"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 512, 230, seed=9)
    X_rows = np.ascontiguousarray(p["X"].T).astype(np.float32)
    W_rows = np.ascontiguousarray(p["A"].T).astype(np.float32)
    offs = [0, 100, 130, 230]
    H, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=150, eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=1e-3, utt_offsets=offs,
                                    info=True, fused_w=4)
    assert info["kernel"] == "k_fused_wide" and info["launches"] >= 2
    for u in range(3):
        a, b = offs[u], offs[u + 1]
        act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X_rows[a:b].astype(np.float64), W_rows.astype(np.float64), 150, 1e-3)
        assert int(info["n_iter"][u]) == n_ref, (u, info["n_iter"], n_ref)
        check(H[a:b], act, rtol=5e-3)
    act, _, _ = o.sklearn_mu_fixed_dictionary(X_rows.astype(np.float64), W_rows.astype(np.float64), 40, 0.0, l1_reg=2.01)
    Hl = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=40, eps_mode="zero_replace", init="sklearn",
                               l1=2.01, fused_w=8)
    check(Hl, act)


def test_wide_stopped_frame_groups_keep_step_with_the_others():
    """Round 4, tagged hand-offs on the static schedule: the wavefronts of a frame group whose utterances have all stopped
    republish their partial sums without sweeping, and must still wait for the previous iteration's sums - without that
    the members of such a group ran ahead of each other's reduce slices, the bounded polls ran out and the solve was redone
    on the two contractions (found by tools/soak_wide.py).  Four utterances of 8 frame tiles each = one frame group each;
    two are a scaled dictionary atom plus a little of another (they converge at the first checks), two are ordinary."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 768, 512, seed=31)
    W_rows = np.ascontiguousarray(p["A"].T).astype(np.float32)
    X_rows = np.ascontiguousarray(p["X"].T).astype(np.float32)
    for a, atom in ((128, 5), (384, 77)):
        X_rows[a:a + 128] = (0.7 * W_rows[atom] + 0.05 * W_rows[atom + 1])[None, :]
    offs = [0, 128, 256, 384, 512]
    kw = dict(layout="frame_major", iters=120, eps_mode="zero_replace", init="sklearn", check_every=5, stop_rule="sklearn",
              tol=2e-3, utt_offsets=offs, info=True)
    H, info = evc.solve_activations(W_rows, X_rows, fused_w=8, **kw)       # (32 frame tiles: below the routing's 43)
    Hr, info_r = evc.solve_activations(W_rows, X_rows, fused=False, **kw)
    assert info["kernel"] == "k_fused_wide" and info["redo"] == 0 and info["members"] > 4, info
    assert list(info["n_iter"]) == list(info_r["n_iter"]), (info["n_iter"], info_r["n_iter"])
    assert int(min(info["n_iter"])) < int(max(info["n_iter"])), info["n_iter"]       # somebody stopped early
    np.testing.assert_allclose(H, Hr, rtol=2e-3, atol=2e-6 * float(Hr.max()))


def test_wide_kl():
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 384, 70, seed=2)
    X_rows = np.ascontiguousarray(p["X"].T).astype(np.float32)
    W_rows = np.ascontiguousarray(p["A"].T).astype(np.float32)
    act, _, _ = o.sklearn_mu_fixed_dictionary_kl(X_rows.astype(np.float64), W_rows.astype(np.float64), 40, 0.0)
    for c in (0, 3, 6):
        Hk, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=40, eps_mode="zero_replace",
                                         init="sklearn", loss="kl", fused_c=c, fused_w=4, info=True)
        assert info["kernel"] == "k_fused_wide"
        check(Hk, act, rtol=5e-3)


def test_wide_repeatable_and_independent_of_the_batch():
    """the partial sums are combined in range order: two runs agree bitwise; an utterance inside a batch equals the
    same utterance alone when both use the same number of ranges (same summation order)"""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 1024, 300, seed=12)
    A32, X32 = p["A"].astype(np.float32), p["X"].astype(np.float32)
    kw = dict(iters=25, eps_mode="zero_replace", init="sklearn", fused_c=5, fused_w=4)
    a = evc.solve_activations(A32, X32, utt_offsets=[0, 120, 300], **kw)
    b = evc.solve_activations(A32, X32, utt_offsets=[0, 120, 300], **kw)
    assert np.array_equal(a, b)
    lone = evc.solve_activations(A32, X32[:, 128:300], iters=25, eps_mode="zero_replace", init="const",
                                 init_value=float(np.sqrt(X32[:, 120:300].astype(np.float64).mean() / 1024)), fused_c=5,
                                 fused_w=4)
    # frames 128.. of the batch sit in whole tiles of utterance 2 and start from the same constant
    np.testing.assert_allclose(a[:, 128:300], lone, rtol=1e-6, atol=0)


def test_wide_real_audio_golden():
    """the |Re STFT| fixture made from the reference's own audio with the installed scikit-learn (tools/make_golden.py),
    float32 as the script's flow is, through this kernel: same stop iteration, H within float32 drift"""
    import exemplars_vc_amd as evc
    g = load_golden(GOLDEN + "/sklearn_audio_stft.npz")
    X, W = g["X_rows"].astype(np.float32), g["W_rows"].astype(np.float32)
    H, info = evc.solve_activations(W, X, layout="frame_major", iters=int(g["max_iter"]), eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=float(g["tol"]), fused_w=4,
                                    info=True)
    assert info["kernel"] == "k_fused_wide" and H.dtype == np.float32
    assert int(info["n_iter"][0]) == int(g["n_iter"])
    check(H.T, g["H"], rtol=2e-2)


def test_routing_by_batch_size():
    """fewer frames than one utterance (43 frame tiles): the two-contraction path; from one utterance on: the fused kernel
    (round 4: static schedule with tagged hand-offs).  M <= 32 is not this kernel's."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(201, 256, 4800, seed=1)
    A32, X32 = p["A"].astype(np.float32), p["X"].astype(np.float32)
    kw = dict(iters=3, eps_mode="zero_replace", init="sklearn", info=True)
    assert evc.solve_activations(A32, X32[:, :400], **kw)[1]["kernel"] == "k_gemm2"       # 25 frame tiles
    assert evc.solve_activations(A32, X32[:, :688], **kw)[1]["kernel"] == "k_fused_wide"  # one utterance (round 4)
    assert evc.solve_activations(A32, X32, **kw)[1]["kernel"] == "k_fused_wide"
    assert evc.solve_activations(A32, X32, fused=False, **kw)[1]["kernel"] == "k_gemm2"
    assert evc.solve_activations(p["A"], p["X"], **kw)[1]["kernel"] == "k_fused_wide64"     # (round 4: 176 < M <= 208 in float64)
    assert evc.solve_activations(p["A"][:150], p["X"][:150], **kw)[1]["kernel"] == "k_gemm_nt"
    assert evc.solve_activations(A32[:25], X32[:25], **kw)[1]["kernel"].startswith("k_fused_")


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_default_flow_at_dictionary_scale_real_audio(tag):
    """The script's default flow at the size it runs it (SURVEY 8d real-audio variant; tools/make_golden_audio.py):
    |Re STFT| of the reference's eight parallel SF1/TF1 utterances, DTW-aligned into a 4096-exemplar dictionary pair,
    wav/SF1_100162.wav (688 frames) converted; expected values from the installed scikit-learn through the call of
    04_align_n_nmf.py:212-213 (tol 1e-4: it stops after 140 iterations), in float32 and in float64.  Through the
    drop-in surface: compat.factorize.factorize(use_stft=True) + convert()."""
    import warnings
    from exemplars_vc_amd.compat.factorize import factorize, convert
    g32 = load_golden(GOLDEN + "/audio_stft_n4096_f32.npz")
    g = g32 if tag == "f32" else load_golden(GOLDEN + "/audio_stft_n4096_f64.npz")
    dt = np.float32 if tag == "f32" else np.float64
    A, B, X = (g32[k].astype(dt) for k in ("A_rows", "B_rows", "X_rows"))
    assert A.shape == (4096, 201) and X.shape == (688, 201)
    # the dictionary as the script holds it: per-file lists of aligned frames (here: four "files" of 1024 pairs)
    src = [{"real": A[i:i + 1024]} for i in range(0, 4096, 1024)]
    tar = [{"real": B[i:i + 1024]} for i in range(0, 4096, 1024)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H, R = factorize({"real": X}, src, use_stft=True, tol=float(g["tol"]))
        Y = convert(H, tar, R, use_stft=True)
    assert R is None and H["H_stft"].shape == (4096, 688) and H["H_stft"].dtype == dt and Y.shape == (688, 201)
    if tag == "f64":
        np.testing.assert_allclose(Y, g["Y_rows"], rtol=1e-8, atol=1e-12 * float(g["Y_rows"].max()))
        np.testing.assert_allclose(H["H_stft"][:, :32], g["H_first32"], rtol=1e-7, atol=1e-14 * float(g["H_first32"].max()))
    else:
        # float32 trajectories of different summation orders drift: 140 iterations over N = 4096
        np.testing.assert_allclose(Y, g["Y_rows"], rtol=2e-3, atol=1e-5 * float(g["Y_rows"].max()))
        np.testing.assert_allclose(H["H_stft"][:, :32], g["H_first32"], rtol=2e-2, atol=1e-5 * float(g["H_first32"].max()))
    # the same utterance through the solver with the stop rule visible: it stops where scikit-learn stopped
    import exemplars_vc_amd as evc
    _, info = evc.solve_activations(A, X, layout="frame_major", iters=int(g["max_iter"]), eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=float(g["tol"]), info=True)
    assert int(info["n_iter"][0]) == int(g["n_iter"]) == 140


@pytest.mark.parametrize("seed", range(16))
def test_wide_random_shapes_against_the_two_contraction_path(seed):
    """Differential test over seeded random shapes: the fused kernel (random ranges per group and wavefronts per
    workgroup) against the two-contraction float32 path on the same call - ragged utterances, exemplar counts that
    are not multiples of 16, both layouts, with the synthesis, error traces over several launches, KL."""
    import exemplars_vc_amd as evc
    o = oracle()
    rng = np.random.default_rng(500 + seed)
    M = int(rng.integers(33, 209))
    N = int(rng.choice([int(rng.integers(40, 300)), int(rng.integers(300, 1500)), 16 * int(rng.integers(8, 80))]))
    lens = [int(rng.integers(1, 200)) for _ in range(int(rng.integers(1, 6)))]
    T = sum(lens)
    p = o.synth_problem(M, N, T, seed=900 + seed)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    K = int(rng.integers(1, 20))
    layout = "frame_major" if seed % 2 else "bin_major"
    tr = (lambda a: np.ascontiguousarray(a.T.astype(np.float32))) if layout == "frame_major" else \
        (lambda a: np.ascontiguousarray(a.astype(np.float32)))
    kw = dict(layout=layout, iters=K, eps_mode=["zero_replace", "add", "clamp"][seed % 3],
              init=["sklearn", "const"][seed % 2], utt_offsets=offs)
    if kw["init"] == "const":
        kw["init_value"] = 0.21
    if seed % 4 == 3:
        kw.update(check_every=4, info=True)
    if seed % 5 == 2:
        kw.update(loss="kl", eps_mode="zero_replace")
    c, w = int(rng.integers(0, 9)), int(rng.choice([4, 8]))
    got = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused_c=c, fused_w=w, **kw)
    want = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused=False, **kw)
    for g, wv, name in zip(got[:2], want[:2], ("H", "Y")):
        g64, w64 = np.asarray(g, dtype=np.float64), np.asarray(wv, dtype=np.float64)
        np.testing.assert_allclose(g64, w64, rtol=2e-3, atol=2e-6 * float(np.abs(w64).max()),
                                   err_msg=f"seed {seed} M={M} N={N} lens={lens} K={K} c={c} w={w} {layout}: {name}")
    if seed % 4 == 3:
        assert got[2]["kernel"] == "k_fused_wide" and want[2]["kernel"] == "k_gemm2"
        np.testing.assert_allclose(got[2]["err"], want[2]["err"], rtol=1e-3, atol=1e-5 * float(np.linalg.norm(p["X"])),
                                   equal_nan=True)


def test_wide_two_processes_share_the_card(tmp_path):
    """The task queue needs no co-residency: two processes on one GPU, each running solves whose tasks wait on each
    other's partial sums, both finish with correct results (a task only ever waits for tasks that hold earlier
    tickets, i.e. that are running)."""
    import os
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        "import exemplars_vc_amd as evc\n"
        "from oracle import evc_oracle as o\n"
        "p = o.synth_problem(201, 1024, 688, seed=int(sys.argv[1]))\n"
        "A, X = p['A'].astype(np.float32), p['X'].astype(np.float32)\n"
        "want = evc.solve_activations(A, X, iters=20, eps_mode='zero_replace', init='sklearn', fused=False)\n"
        "worst = 0.0\n"
        "for i in range(3):\n"
        "    got, info = evc.solve_activations(A, X, iters=20, eps_mode='zero_replace', init='sklearn', fused_w=4, info=True)\n"
        "    assert info['kernel'] == 'k_fused_wide' and info['members'] > 4\n"
        "    worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))\n"
        "print('WORST', worst)\n"
        "assert worst < 1e-4, worst\n")
    procs = [subprocess.Popen([sys.executable, str(script), str(k)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for k in (1, 2)]
    outs = [pr.communicate(timeout=600)[0].decode() for pr in procs]
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0 and "WORST" in out, out[-2000:]
