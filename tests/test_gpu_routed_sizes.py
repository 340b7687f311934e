"""The kernels at the sizes they are ROUTED for, default routing, no tuning bits - and what the
library does when a task-queue solve is voided (item 2 vii).

k_fused_wide64 serves float64 batches of 240 ... 1000 frame tiles of 208 < M <= 528 bins (6 to ~20 utterances),
k_fused_wide float32 batches from ~300 frame tiles on of 32 < M <= 208 bins; the tests of test_gpu_wide*.py force them
on small problems through the tuning bits.  Here: BASELINE's C3 shape (M = 513, N = 8192, K = 200) as a batch of 16
utterances and as one call of 16 384 frames, the STFT flow (M = 201, N = 4096, K = 150, float32) as a batch of 16, and
the N = 4096 real-audio golden replicated 16 times through the drop-in's batch entry.  A frame column evolves
independently of the others, so a slice of frames solved by the oracle from the batch's start value must equal the
same frames of the batch; the utterances of a batch are copies of one, so every copy must equal the first.
Tolerances: float64 1e-8 pure relative; float32 as tests/test_gpu_wide.py (2e-3 / floor 1e-6 of the maximum on short
runs), with the measured drift at K = 150, N = 4096 printed and bounded."""
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_err

pytestmark = pytest.mark.gpu
RTOL64 = 1e-8


def oracle():
    from oracle import evc_oracle as o
    return o


def _batch_of_copies(p, n_utt, dtype=np.float64):
    """frames-as-rows batch: n_utt copies of the problem's utterance"""
    X1 = np.ascontiguousarray(p["X"].T).astype(dtype)
    A = np.ascontiguousarray(p["A"].T).astype(dtype)
    B = np.ascontiguousarray(p["B"].T).astype(dtype)
    T = X1.shape[0]
    return A, B, np.ascontiguousarray(np.tile(X1, (n_utt, 1))), np.arange(n_utt + 1, dtype=np.int32) * T


def _slice_want(o, A_rows, X_rows, K, frames, h0, eps=None):
    A64, X64 = A_rows.astype(np.float64), X_rows.astype(np.float64)
    N = A64.shape[0]
    return o.mu_solve(np.ascontiguousarray(A64.T), np.ascontiguousarray(X64[:frames].T), np.full((N, frames), h0), K,
                      eps_mode=o.EPS_ZERO_REPLACE, eps=o.SK_EPSILON if eps is None else eps, l1=0.0, algo="factored")


def test_c3_batch_of_16_utterances_on_the_fused_float64_kernel():
    """M = 513, N = 8192, K = 200, 16 x 688 frames: 688 frame tiles -> k_fused_wide64 by default"""
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, K, T, U = 513, 8192, 200, 688, 16
    p = o.synth_problem(M, N, T, seed=513)
    A, B, X, offs = _batch_of_copies(p, U)
    dev = torch.device("cuda:0")
    At, Bt, Xt = (torch.from_numpy(a).to(dev) for a in (A, B, X))
    H, Y, info = evc.convert(At, Xt, Bt, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                             check_every=20, utt_offsets=offs, info=True)
    assert info["kernel"] == "k_fused_wide64" and info["redo"] == 0, info
    assert tuple(H.shape) == (U * T, N) and bool(torch.isfinite(H).all()) and bool((H >= 0).all())
    assert (info["n_iter"] == K).all()
    for u in range(U):
        tr = info["err"][u, 1:]
        assert np.all(np.diff(tr) <= 1e-12 * tr[0]), "the Frobenius residual must not increase under MU"
    h0 = float(np.sqrt(X[:T].mean() / N))
    want = _slice_want(o, A, X, K, 64, h0)
    got = H[:64].cpu().numpy().T
    r, z = rel_err(got, want)
    assert r <= RTOL64 and z == 0.0, f"C3 x16 slice: max rel err {r:.3e}"
    np.testing.assert_allclose(info["err"][0, -1], o.residual_fro(p["A"], p["X"], H[:T].cpu().numpy().T), rtol=1e-9)
    # every utterance is a copy of the first (other frame groups, other positions in the task queue)
    H0 = H[:T]
    for u in (1, 7, 15):
        assert float((H[u * T:(u + 1) * T] - H0).abs().max() / H0.abs().max()) <= 1e-12
    Yw = (H[:128].double() @ Bt.double()).cpu().numpy()
    np.testing.assert_allclose(Y[:128].cpu().numpy(), Yw, rtol=1e-10, atol=1e-14 * float(np.abs(Yw).max()))


@pytest.mark.parametrize("N,U,kernel", [(8192, 3, "k_fused_wide64"), (8192, 2, "k_gemm_nt"), (1024, 1, "k_fused_wide64"),
                                        (1024, 2, "k_fused_wide64")])
def test_c3_width_small_batches_by_the_round_4_routing(N, U, kernel):
    """Round 4's routing table (profiles/r04_routing_table.md): at M = 513 three utterances go to k_fused_wide64 (260 tasks
    through the ticket queue), two stay on the two contractions; small dictionaries run the fused kernel from one utterance
    on (the static schedule: one sweep task per workgroup).  A 48-frame slice against the float64 oracle each."""
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    M, K, T = 513, 60, 688
    p = o.synth_problem(M, N, T, seed=N + U)
    A, B, X, offs = _batch_of_copies(p, U)
    dev = torch.device("cuda:0")
    At, Xt = torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev)
    H, info = evc.solve_activations(At, Xt, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                    utt_offsets=offs, info=True)
    assert info["kernel"] == kernel and info["redo"] == 0, info
    want = _slice_want(o, A, X, K, 48, float(np.sqrt(X[:T].mean() / N)))
    r, z = rel_err(H[:48].cpu().numpy().T, want)
    assert r <= RTOL64 and z == 0.0, f"M=513 N={N} x{U} slice ({info['kernel']}): max rel err {r:.3e}"
    if U > 1:
        assert float((H[(U - 1) * T:(U - 1) * T + 64] - H[:64]).abs().max() / H[:64].abs().max()) <= 1e-12


def test_c3_one_call_of_16384_frames():
    """BASELINE.md C3, "T = 16 384": 1024 frame tiles - beyond k_fused_wide64's window, the two contractions serve it"""
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, K, T = 513, 8192, 200, 16384
    p = o.synth_problem(M, N, 688, seed=514)
    A, B, X, _ = _batch_of_copies(p, 24)
    X = np.ascontiguousarray(X[:T])
    dev = torch.device("cuda:0")
    At, Xt = torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev)
    H, info = evc.solve_activations(At, Xt, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                    info=True)
    assert info["kernel"] in ("k_gemm_nt", "k_fused_wide64") and info["redo"] == 0, info
    assert tuple(H.shape) == (T, N) and bool(torch.isfinite(H).all())
    want = _slice_want(o, A, X, K, 64, float(np.sqrt(X.mean() / N)))
    r, z = rel_err(H[:64].cpu().numpy().T, want)
    assert r <= RTOL64 and z == 0.0, f"C3 T=16384 slice ({info['kernel']}): max rel err {r:.3e}"
    # frames 688.. repeat frames 0..: same start value, same columns
    assert float((H[688:688 + 64] - H[:64]).abs().max() / H[:64].abs().max()) <= 1e-12


@pytest.mark.parametrize("U", [16, 2, 5])
def test_stft_flow_batches_on_the_fused_float32_kernel(U):
    """M = 201, N = 4096, K = 150, float32, U x 688 frames -> k_fused_wide (16: the task queue; 2 and 5: the static
    schedule of round 4, one sweep task per workgroup); 48 frames against the float64 oracle on the
    same float32 inputs.  The drift of a float32 trajectory over 150 iterations of 4096-term sums is reported and bounded
    (north_star's 1e-4 is asked of the float64 path; scikit-learn's own float32 run differs from its float64 run by the
    same order, tests/golden/audio_stft_n4096_f32 vs _f64)."""
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, K, T = 201, 4096, 150, 688
    p = o.synth_problem(M, N, T, seed=201)
    A, B, X, offs = _batch_of_copies(p, U, np.float32)
    dev = torch.device("cuda:0")
    At, Xt = torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev)
    H, info = evc.solve_activations(At, Xt, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                    utt_offsets=offs, info=True)
    assert info["kernel"] == "k_fused_wide" and info["redo"] == 0 and H.dtype == torch.float32, info
    h0 = float(np.sqrt(X[:T].astype(np.float64).mean() / N))
    want = _slice_want(o, A, X, K, 48, h0)
    got = H[:48].cpu().numpy().astype(np.float64).T
    big = want > 1e-4 * want.max()
    drift_big = float(np.max(np.abs(got[big] - want[big]) / want[big]))
    drift_abs = float(np.max(np.abs(got - want)) / want.max())
    print(f"k_fused_wide at K=150 N=4096: max rel drift on entries > 1e-4 max: {drift_big:.2e}; max abs / max: {drift_abs:.2e}")
    assert drift_big <= 2e-3 and drift_abs <= 1e-4
    H0 = H[:T]
    for u in (1, U // 2, U - 1):
        assert float((H[u * T:(u + 1) * T] - H0).abs().max() / H0.abs().max()) <= 1e-5


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_real_audio_golden_replicated_16_times_through_the_batch_entry(tag):
    """tests/golden/audio_stft_n4096_*: the reference's audio, scikit-learn's own output (140 iterations at tol 1e-4).
    Sixteen copies through compat.factorize_utterances (one launch sequence): every copy stops where scikit-learn
    stopped and synthesises scikit-learn's Y; the fused float32 kernel serves the float32 batch, k_fused_wide64 (3 whole
    bin tiles per wavefront + the split one, round 4) the float64 batch - at 1e-8 of scikit-learn's output."""
    from exemplars_vc_amd.compat.factorize import factorize_utterances, synthesize_rows
    g32 = load_golden(GOLDEN + "/audio_stft_n4096_f32.npz")
    g = g32 if tag == "f32" else load_golden(GOLDEN + "/audio_stft_n4096_f64.npz")
    dt = np.float32 if tag == "f32" else np.float64
    A, B, X = (g32[k].astype(dt) for k in ("A_rows", "B_rows", "X_rows"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Hs, n_iter, info = factorize_utterances([X] * 16, A, tol=float(g["tol"]), return_info=True)
    assert info["kernel"] == ("k_fused_wide" if tag == "f32" else "k_fused_wide64"), info
    assert [int(n) for n in n_iter] == [int(g["n_iter"])] * 16 == [140] * 16
    for u in (0, 5, 15):
        Y = synthesize_rows(Hs[u], B)
        if tag == "f64":
            np.testing.assert_allclose(Y, g["Y_rows"], rtol=1e-8, atol=1e-12 * float(g["Y_rows"].max()))
            np.testing.assert_allclose(Hs[u][:, :32], g["H_first32"], rtol=1e-7, atol=1e-14 * float(g["H_first32"].max()))
        else:
            np.testing.assert_allclose(Y, g["Y_rows"], rtol=2e-3, atol=1e-5 * float(g["Y_rows"].max()))
            np.testing.assert_allclose(Hs[u][:, :32], g["H_first32"], rtol=2e-2, atol=1e-5 * float(g["H_first32"].max()))


@pytest.mark.parametrize("dtype,M,kernel,redo_kernel", [(np.float32, 201, "k_fused_wide", "k_gemm2"),
                                                         (np.float64, 513, "k_fused_wide64", "k_gemm_nt")])
@pytest.mark.parametrize("when", [True, 2])
def test_a_voided_task_queue_solve_is_redone_and_says_so(dtype, M, kernel, redo_kernel, when):
    """A bounded wait of the task-queue kernels that runs out raises a flag (evc_wide.hip / evc_wide64.hip).  Round 3
    exported NaN under status 0; now the flag is read back before anything reaches H or Y and the solve is redone on the
    two contractions (evc_solve_info.redo = 1).  The test raises the flag as a timed-out wait would
    (evc_solve_opts.test_abort_at): at the start of the call, and in front of a later launch of a checked solve."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(M, 384, 150, seed=3)
    A, X, B = p["A"].astype(dtype), p["X"].astype(dtype), p["B"].astype(dtype)
    H0 = (np.random.default_rng(1).random((384, 150)) + 1e-4).astype(dtype)
    # (60 iterations, a check every 4: three launches on either kernel - k_fused_wide takes up to five checks per launch)
    kw = dict(iters=60, eps_mode="add", eps=1e-9, fused_w=4, check_every=4, info=True)
    Hc, Yc, ic = evc.convert(A, X, B, H0.copy(), **kw)
    assert ic["kernel"] == kernel and ic["redo"] == 0, ic
    Hr, Yr, ir = evc.convert(A, X, B, H0.copy(), _fake_coop_timeout=when, **kw)
    assert ir["redo"] == 1 and ir["kernel"] == redo_kernel, ir
    assert np.isfinite(Hr).all() and np.isfinite(Yr).all()
    want = o.mu_solve(A.astype(np.float64), X.astype(np.float64), H0.astype(np.float64), 60, eps_mode=o.EPS_ADD, eps=1e-9,
                      algo="factored")
    tol = dict(rtol=1e-8, atol=0) if dtype == np.float64 else dict(rtol=2e-3, atol=1e-6 * float(want.max()))
    np.testing.assert_allclose(Hr, want, **tol)
    np.testing.assert_allclose(Hc, want, **tol)
    np.testing.assert_allclose(Yr, B.astype(np.float64) @ want, rtol=tol["rtol"], atol=1e-6 * float((B @ want).max()))


def test_no_exchange_flag_and_tuning_values():
    """EVC_FLAG_NO_EXCHANGE (cooperative=False) rules the task queues out as it rules k_fused_all out; tuning values
    outside the instances that exist are an error, not a silent fallback"""
    import exemplars_vc_amd as evc
    from exemplars_vc_amd._lib import EvcError
    o = oracle()
    p = o.synth_problem(201, 256, 4800, seed=1)
    A32, X32 = p["A"].astype(np.float32), p["X"].astype(np.float32)
    kw = dict(iters=3, eps_mode="zero_replace", init="sklearn", info=True)
    assert evc.solve_activations(A32, X32, **kw)[1]["kernel"] == "k_fused_wide"
    assert evc.solve_activations(A32, X32, cooperative=False, **kw)[1]["kernel"] == "k_gemm2"
    with pytest.raises(EvcError) as e:
        evc.solve_activations(A32, X32, fused_w=5, **kw)
    assert e.value.status == -1
    p = o.synth_problem(513, 256, 64, seed=2)
    with pytest.raises(EvcError) as e:
        evc.solve_activations(p["A"], p["X"], fused_w=9, **kw)
    assert e.value.status == -1
    assert evc.solve_activations(p["A"], p["X"], fused_w=4, cooperative=False, **kw)[1]["kernel"] == "k_gemm_nt"


def test_world_flow_warnings_come_from_the_callers_thread():
    """compat.factorize (WORLD branch) solves sp / ap / f0 on three host threads and three streams.  The solves' warnings
    are issued by the calling thread after the join; no worker enters warnings.catch_warnings, whose save / restore of
    the process-global filter list breaks when exits interleave: afterwards a warning is still shown and the
    filter list is what it was."""
    from exemplars_vc_amd.compat.factorize import factorize
    try:
        from sklearn.exceptions import ConvergenceWarning
    except Exception:  # pragma: no cover
        ConvergenceWarning = UserWarning
    rng = np.random.default_rng(7)
    src = [{"sp": rng.random((300, 513)) + 0.05, "ap": rng.random((300, 513)) + 0.05, "f0": rng.random(300) + 0.05}
           for _ in range(2)]
    utt = {"sp": rng.random((90, 513)) + 0.05, "ap": rng.random((90, 513)) + 0.05, "f0": rng.random(90) + 0.05}
    before = list(warnings.filters)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        inner = list(warnings.filters)
        H, R = factorize(utt, src, use_stft=False, tol=1e-12)      # tol so small that every stream runs to max_iter
        assert list(warnings.filters) == inner, "a worker thread left its filter list behind"
        warnings.warn("still shown", UserWarning)
    assert list(warnings.filters) == before
    cats = [w.category for w in rec]
    # (sp and ap run to max_iter; the one-bin f0 stream fits exactly and stops on its own)
    assert sum(c is ConvergenceWarning for c in cats) >= 2, cats
    assert any(str(w.message) == "still shown" for w in rec)
    assert set(H) == {"H_sp", "H_ap", "H_f0"} and H["H_sp"].shape == (600, 90)


def test_pair_tiles_kernel_parity():
    """k_fused_xy (evc_fused_xy.hip; EVC_FLAG_PAIR_TILES, not routed to by default: profiles/r04_xy_notes.md): parity
    with the oracle on member counts 2 ... 16, lone tiles, L1 through the spare bin and through registers"""
    import exemplars_vc_amd as evc
    o = oracle()
    for M, N, T, K, l1 in ((25, 512, 64, 10, 0.0), (25, 768, 33, 10, 0.0), (25, 4096, 200, 20, 0.25), (32, 1024, 64, 8, 0.1),
                           (13, 1024, 70, 8, 0.05), (7, 2048, 50, 6, 0.0), (25, 3000, 100, 10, 0.0)):
        p = o.synth_problem(M, N, T, seed=M + N + T)
        H, info = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", l1=l1,
                                        pair_tiles=True, info=True)
        assert info["kernel"] == "k_fused_xy" and info["redo"] == 0, info
        want = o.mu_solve(p["A"], p["X"], np.full((N, T), np.sqrt(p["X"].mean() / N)), K, eps_mode=o.EPS_ZERO_REPLACE,
                          eps=o.SK_EPSILON, l1=l1, algo="factored")
        r, z = rel_err(H, want)
        assert r <= RTOL64 and z == 0.0, (M, N, T, r)
