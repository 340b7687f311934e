"""Every BASELINE.json configuration at its stated size on the HIP path (`-m gpu`).

C1  M=25  N=512   K=50   T=688            complete comparison with the oracle
C2  M=25  N=4096  K=100                   tests/test_gpu_parity.py::test_c2_full_size_properties (+ bench parity)
C3  M=513 N=8192  K=200  T=688            64-frame slice entry by entry + properties of the whole batch
C4  the 162-utterance set, N=4096, K<=100, tol=1e-4, through shard.convert_sharded on one GPU
C5  N=16384 + L1 (sklearn l1_reg_W = M*0.01), K=100, T=688, M=25 and M=513

The oracle is float64 numpy; a frame column evolves independently of the others given the dictionary, so a
slice of frames solved by the oracle alone (same constant start value as the batch: sklearn's
sqrt(mean(X_utterance)/N)) must equal the same frames of the batch.  The FACTORED oracle costs 4MNK flop per
frame, which keeps a 64-frame slice at a few seconds even for C3.  Tolerance: RTOL64 = 1e-8 pure relative
(north_star asks for 1e-4).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

RTOL64 = 1e-8
SLICE = 64
# utterance lengths of the audio bundled with the reference, cycled to the corpus size
C4_LENGTHS = [704, 216, 513, 494, 945, 640, 497, 1370, 688]


def oracle():
    from oracle import evc_oracle
    return evc_oracle


def assert_close64(got, want, what, rtol=RTOL64):
    r, z = rel_err(got, want)
    assert r <= rtol and z == 0.0, f"{what}: max rel err {r:.3e}, max |got| where want==0 {z:.3e}"


def _slice_check(p, H, N, K, l1, what, frames=SLICE):
    """frames 0..frames-1 of the batch against the oracle started from the batch's init value"""
    o = oracle()
    h0 = np.sqrt(p["X"].mean() / N)
    want = o.mu_solve(p["A"], p["X"][:, :frames], np.full((N, frames), h0), K, eps_mode=o.EPS_ZERO_REPLACE,
                      eps=o.SK_EPSILON, l1=l1, algo="factored")
    assert_close64(H[:, :frames], want, what)


def test_c1_full_size():
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, K, T = 25, 512, 50, 688
    p = o.synth_problem(M, N, T, seed=20190131)
    H, Y = evc.convert(p["A"], p["X"], p["B"], iters=K, eps_mode="zero_replace", init="sklearn")
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary(np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T),
                                                   max_iter=K, tol=0.0)
    assert_close64(H, act.T, "C1 H (whole utterance, scikit-learn's Gram algebra)")
    assert_close64(Y, p["B"] @ act.T, "C1 Y")
    for kw in (dict(all_resident=False), dict(algo="gram"), dict(fused=False)):
        assert_close64(evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", **kw),
                       act.T, f"C1 {kw}")


def test_c3_full_size():
    """M=513 bins, N=8192, K=200: the generic path (two contractions per iteration, split-K for the
    skinny one), one 688-frame utterance."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, N, K, T = 513, 8192, 200, 688
    p = o.synth_problem(M, N, T, seed=513)
    H, info = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn",
                                    check_every=20, info=True)
    assert H.shape == (N, T) and np.isfinite(H).all() and (H >= 0).all()
    assert int(info["n_iter"][0]) == K
    tr = info["err"][0, 1:]          # (slot 0, the error at init, is only evaluated for the sklearn stop rule)
    assert np.all(np.diff(tr) <= 1e-12 * tr[0]), "the Frobenius residual must not increase under MU"
    np.testing.assert_allclose(tr[-1], o.residual_fro(p["A"], p["X"], H), rtol=1e-9)
    _slice_check(p, H, N, K, 0.0, "C3 slice")
    Y = evc.synthesize(p["B"], H)
    assert_close64(Y, p["B"] @ H, "C3 Y = B H", rtol=1e-10)


@pytest.mark.parametrize("M", [25, 513])
def test_c5_full_size(M):
    """N=16384 exemplars with sklearn's L1 penalty (l1_reg_W = n_features * alpha_W * l1_ratio with
    alpha_W=0.01, l1_ratio=1), K=100, one 688-frame utterance; M=25 (fused path: 87 % of the tiles would
    stream through HBM without the all-resident kernel) and M=513 (generic path)."""
    import exemplars_vc_amd as evc
    o = oracle()
    N, K, T = 16384, 100, 688
    l1 = M * 0.01
    p = o.synth_problem(M, N, T, seed=16384 + M)
    H = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", l1=l1)
    assert H.shape == (N, T) and np.isfinite(H).all() and (H >= 0).all()
    _slice_check(p, H, N, K, l1, f"C5 M={M} slice")
    if M == 25:
        # the same frames inside a 16-utterance batch (one workgroup pair per CU walks many tiles) and on the
        # kernels without inter-workgroup exchange
        reps = 4
        Xb = np.concatenate([p["X"]] * reps, axis=1)
        offs = [T * i for i in range(reps + 1)]
        Hb = evc.solve_activations(p["A"], Xb, iters=K, eps_mode="zero_replace", init="sklearn", l1=l1,
                                   utt_offsets=offs)
        for i in range(reps):
            assert_close64(Hb[:, i * T:(i + 1) * T], H, f"C5 batch copy {i}", rtol=1e-10)
        Hn = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", l1=l1,
                                   cooperative=False)
        assert_close64(Hn, H, "C5 without exchange", rtol=1e-10)


@pytest.mark.parametrize("M,N", [(25, 8192), (25, 32768), (13, 16384), (32, 8192), (4, 16384),
                                 (25, 1536), (32, 1536), (25, 3000), (25, 5000), (17, 6100), (25, 21000), (32, 40000)])
def test_reduce_scatter_exchange_member_counts(M, N):
    """k_fused_all with run-time member counts (reduce-scatter exchange): 16, 32 and 64 members, and the counts that
    dictionaries of arbitrary size produce after padding to whole members - 3 (N=1536), 6 (3000), 10 (5000), 12
    (6100), 42 (21000), 79 (40000) - with bins that fill 1, 4, 5, 7 and 8 k-steps (slices of the exchange that are
    ragged or empty): against the oracle, against the kernels without exchange, and twice for bitwise equality
    (every member must obtain the same V' whatever the arrival order)."""
    import exemplars_vc_amd as evc
    o = oracle()
    K, T = 30, 200                        # 13 frame tiles: more groups than tiles at 16 members
    p = o.synth_problem(M, N, T, seed=N + M)
    H = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn")
    _slice_check(p, H, N, K, 0.0, f"M={M} N={N}", frames=32)
    Hn = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn", cooperative=False)
    assert_close64(H, Hn, "against the kernels without exchange", rtol=1e-10)
    H2 = evc.solve_activations(p["A"], p["X"], iters=K, eps_mode="zero_replace", init="sklearn")
    assert np.array_equal(H, H2), "two runs of the exchange must agree bit for bit"


def _c4_problem(n_utt=162, N=4096, M=25, seed=4):
    o = oracle()
    p = o.synth_problem(M, N, 0, seed=seed)
    rng = np.random.default_rng(seed + 1)
    Xs = []
    for i in range(n_utt):
        T = C4_LENGTHS[i % len(C4_LENGTHS)]
        Hs = rng.random((N, T)) * (rng.random((N, T)) < 8.0 / N) * (1.0 + (i % 5))
        Xs.append(np.ascontiguousarray((p["A"] @ Hs + 1e-6).T))
    return p, Xs


def test_c4_full_set_on_one_gpu():
    """The 162-utterance set (1.09e5 frames) through the shard driver with world_size 1: one batched launch
    sequence, the reference's per-call semantics per utterance (own init value, stop test every 10
    iterations with tol=1e-4, own n_iter).  Twelve utterances (three each of 216, 494, 497 and 513 frames, spread over
    the set - the scikit-learn restatement evaluates the Gram algebra, 2 N^2 flop per frame-iteration, so the short
    ones are sampled) are compared with the restatement run on each of them alone: same n_iter, same Y."""
    from exemplars_vc_amd.shard import convert_sharded
    o = oracle()
    p, Xs = _c4_problem()
    assert sum(len(x) for x in Xs) == 18 * sum(C4_LENGTHS)
    W, B = np.ascontiguousarray(p["A"].T), np.ascontiguousarray(p["B"].T)
    out = convert_sharded(Xs, W, B, rank=0, world_size=1, iters=100, tol=1e-4, info=True)
    assert len(out) == len(Xs)
    for i, (Y, n) in enumerate(out):
        assert Y.shape == (len(Xs[i]), 25) and np.isfinite(Y).all() and 10 <= n <= 100 and n % 10 == 0
    by_len = {L: [i for i in range(len(Xs)) if len(Xs[i]) == L] for L in (216, 494, 497, 513)}
    sample = sorted(i for idx in by_len.values() for i in (idx[0], idx[len(idx) // 2], idx[-1]))
    assert len(sample) == 12
    for i in sample:
        act, n_ref, _ = o.sklearn_mu_fixed_dictionary(Xs[i], W, max_iter=100, tol=1e-4)
        assert out[i][1] == n_ref, (i, out[i][1], n_ref)
        assert_close64(out[i][0], act @ B, f"C4 utterance {i}")


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)      # before anything touches the GPU
from test_gpu_configs import _c4_problem
from exemplars_vc_amd.shard import convert_sharded
p, Xs = _c4_problem(n_utt=27, seed=9)
W, B = np.ascontiguousarray(p["A"].T), np.ascontiguousarray(p["B"].T)
res = convert_sharded(Xs, W, B, iters=100, tol=1e-4, info=True, device="cuda:0")
if rank == 0:
    np.savez(out, *[y for y, _ in res], n_iter=np.array([n for _, n in res]))
else:
    assert res is None
dist.barrier()
dist.destroy_process_group()
"""


def test_two_process_shards_with_the_hip_solver(tmp_path):
    """The N>1 path with the real solver: two fresh processes (gloo rendezvous, both on device 0 - a GPU box
    here has one device), each converting its LPT shard of 27 utterances of the C4 length set with
    libevc_hip.so, rank 0 gathering; against the same list converted by one process."""
    import socket
    from exemplars_vc_amd.shard import convert_sharded
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    logs = [pr.communicate(timeout=600)[0].decode() for pr in procs]
    for pr, log in zip(procs, logs):
        assert pr.returncode == 0, log[-3000:]
    got = np.load(out)
    p, Xs = _c4_problem(n_utt=27, seed=9)
    W, B = np.ascontiguousarray(p["A"].T), np.ascontiguousarray(p["B"].T)
    want = convert_sharded(Xs, W, B, rank=0, world_size=1, iters=100, tol=1e-4, info=True)
    assert len(got.files) == len(Xs) + 1
    assert np.array_equal(got["n_iter"], np.array([n for _, n in want]))
    for i, (Y, _) in enumerate(want):
        # a shard is a different batch: frames are independent columns, so only the launch mode (how many
        # workgroups share a frame tile) can differ - summation order of V', 1e-13
        assert_close64(got[f"arr_{i}"], Y, f"utterance {i}", rtol=1e-10)


@pytest.mark.parametrize("N", [1040, 2000])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_unguarded_mode_with_all_padding_exemplar_tiles(N, dt):
    """nmf_tool's unguarded update (eps_mode 'none') where the packed layout holds whole all-zero exemplar
    tiles (N >= 1024 is padded to a multiple of 128): 0 * 0 / 0 there must not leak NaN into V and from
    there into every frame.  float32 rides the float64 fused kernels (M <= 32)."""
    import exemplars_vc_amd as evc
    o = oracle()
    M, T, K = 25, 40, 30
    p = o.synth_problem(M, N, T, seed=N)
    A, X = p["A"].astype(dt), p["X"].astype(dt)
    H0 = (np.random.default_rng(N).random((N, T)) + 1e-4).astype(dt)
    want = o.mu_solve(A.astype(np.float64), X.astype(np.float64), H0.astype(np.float64), K, eps_mode=o.EPS_NONE,
                      eps=0.0, algo="factored")
    got = evc.solve_activations(A, X, H0, iters=K, eps_mode="none", eps=0.0)
    assert got.dtype == dt and np.isfinite(got).all()
    if dt == np.float64:
        assert_close64(got, want, f"none N={N}")
    else:
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-30)


def test_single_frame_at_real_width():
    """The consumer's shape (05_conversion.py:100-106): ONE 513-bin frame against a 513 x N dictionary."""
    import exemplars_vc_amd as evc
    from exemplars_vc_amd.compat.factorize import _factorize
    import warnings
    o = oracle()
    M, N = 513, 2048
    p = o.synth_problem(M, N, 1, seed=5)
    X_rows, W_rows = np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H = _factorize(X_rows, W_rows, tol=1e-4)
    act, n_iter, _ = o.sklearn_mu_fixed_dictionary(X_rows, W_rows, 150, 1e-4)
    assert H.shape == (N, 1)
    assert_close64(H, act.T, "T=1, M=513")
    Y = evc.synthesize(np.ascontiguousarray(p["B"].T), H.T.copy(), layout="frame_major")
    assert_close64(Y, act @ p["B"].T, "T=1 synthesis")


def test_two_host_threads_on_two_streams():
    """include/evc.h: calls on distinct streams are independent.  The Python surface keeps one scratch buffer per
    (device, stream): two host threads, each on its own stream, solve different problems at the same time (generic
    path and fused path - both use the scratch heavily) and must each get the result of a solo run."""
    import threading
    import torch
    import exemplars_vc_amd as evc
    o = oracle()
    probs = [o.synth_problem(201, 512, 300, seed=1), o.synth_problem(25, 1024, 700, seed=2)]
    kw = dict(iters=40, eps_mode="zero_replace", init="sklearn", cooperative=False)
    solo = [evc.solve_activations(p["A"], p["X"], **kw) for p in probs]
    dev = torch.device("cuda")
    res, errs = [[None] * 6, [None] * 6], []

    def work(i):
        try:
            st = torch.cuda.Stream(device=dev)
            A = torch.from_numpy(probs[i]["A"]).to(dev)
            X = torch.from_numpy(probs[i]["X"]).to(dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                for r in range(6):
                    res[i][r] = evc.solve_activations(A, X, **kw)
                st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        for r in range(6):
            assert np.array_equal(res[i][r].cpu().numpy(), solo[i]), (i, r)


@pytest.mark.parametrize("M,N", [(25, 512), (7, 400), (25, 8192), (32, 16384), (13, 21000)])
def test_kl_update_on_the_all_resident_kernel(M, N):
    """The generalised-KL update (sklearn beta_loss='kullback-leibler', update_H=False) on k_fused_all: one member
    (no exchange) and 16, 32 and 42 members (reduce-scatter), start values formed in the kernel, an all-zero frame
    (numerator 0), with and without the error trace - against the oracle's restatement of scikit-learn and
    against the streamed kernel."""
    import exemplars_vc_amd as evc
    o = oracle()
    K, T = 25, 90
    p = o.synth_problem(M, N, T, seed=7 * N + M)
    X = np.ascontiguousarray(p["X"].T)
    X[5] = 0.0
    W = np.ascontiguousarray(p["A"].T)
    want, n_iter, _ = o.sklearn_mu_fixed_dictionary_kl(X, W, K, 0.0)
    kw = dict(layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn", loss="kl")
    got = evc.solve_activations(W, X, **kw)
    assert_close64(got, want, f"KL all-resident M={M} N={N}")
    res = evc.solve_activations(W, X, all_resident=False, **kw)
    assert_close64(got, res, "against the streamed kernel", rtol=1e-10)
    got2, info = evc.solve_activations(W, X, check_every=5, info=True, **kw)
    res2, info_r = evc.solve_activations(W, X, check_every=5, info=True, all_resident=False, **kw)
    assert_close64(got2, want, "KL with error trace")
    np.testing.assert_allclose(info["err"], info_r["err"], rtol=1e-9, equal_nan=True)


def test_the_library_reports_which_kernel_ran():
    """evc_solve_info (include/evc.h): the kernel, the members per frame tile, the launches and whether a redo
    happened come from the library, not from a guess about the shape (missing item 4)."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 4096, 688 * 3, seed=11)
    kw = dict(iters=20, eps_mode="zero_replace", init="sklearn", info=True)
    _, i = evc.solve_activations(p["A"], p["X"], **kw)                      # C2's shape: the all-resident kernel
    assert (i["kernel"], i["members"], i["redo"], i["exchange"], i["launches"]) == ("k_fused_all", 8, 0, 1, 1), i
    _, i = evc.solve_activations(p["A"], p["X"], cooperative=False, **kw)   # no exchange allowed
    assert i["kernel"] in ("k_fused_res", "k_fused_mu") and i["members"] == 1 and i["exchange"] == 0, i
    _, i = evc.solve_activations(p["A"], p["X"], check_every=10, stop_rule="sklearn", tol=1e-9, **kw)
    assert i["launches"] == 2 and i["redo"] == 0, i                         # one launch per 10 iterations
    _, i = evc.solve_activations(p["A"], p["X"], _fake_coop_timeout=True, **kw)
    assert i["redo"] == 1 and i["exchange"] == 0 and i["launches"] == 2, i
    _, i = evc.solve_activations(p["A"], p["X"], fused=False, **kw)
    assert i["kernel"] == "k_gemm_nt" and i["launches"] == 40, i
    q = o.synth_problem(201, 256, 64, seed=3)
    _, i = evc.solve_activations(q["A"].astype(np.float32), q["X"].astype(np.float32), fused=False, **kw)
    assert i["kernel"] == "k_gemm2", i


@pytest.mark.parametrize("loss", ["frobenius", "kl"])
def test_tile_grid_padded_past_the_dictionary_array(loss):
    """N = 3600: the exemplar tile grid is padded to whole 512-exemplar members (4096 slots) while the imported
    dictionary has 3712 rows: the packing must write zeros for the slots beyond the array instead of
    reading past it.  The workspace is pre-filled with NaN bytes so that anything read from beyond shows."""
    import torch
    import exemplars_vc_amd as evc
    from exemplars_vc_amd import solver
    o = oracle()
    M, N, T, K = 25, 3600, 96, 30
    p = o.synth_problem(M, N, T, seed=8)
    nbytes = solver.workspace_bytes(M, N, T, Mb=M)
    dev = torch.device("cuda", 0)
    with solver._workspace(nbytes, dev) as ws:       # poison the scratch buffer the next call will carve
        ws.fill_(0xFF)
    if loss == "kl":
        X_rows, W_rows = np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T)
        want = o.sklearn_mu_fixed_dictionary_kl(X_rows, W_rows, K, 0.0)[0].T
        H, Y = evc.convert(p["A"], p["X"], p["B"], iters=K, eps_mode="zero_replace", init="sklearn", loss="kl")
    else:
        want = o.sklearn_mu_fixed_dictionary(np.ascontiguousarray(p["X"].T), np.ascontiguousarray(p["A"].T), K, 0.0)[0].T
        H, Y = evc.convert(p["A"], p["X"], p["B"], iters=K, eps_mode="zero_replace", init="sklearn")
    assert np.isfinite(H).all() and np.isfinite(Y).all()
    np.testing.assert_allclose(H, want, rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(Y, p["B"] @ want, rtol=1e-8)


def test_bench_gpus_flag_starts_the_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks itself (here both on the one
    card, over gloo) and reports them: n_gpus, ranks_seen, one rate per rank; C4 is one set split over the ranks."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--dist-backend", "gloo",
           "--steps", "2", "--warmup", "1", "--no-cpu", "--no-pcie", "--config", "C4"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and len(r["frames_per_s_per_rank"]) == 2, r
    assert r["scaling"] == "strong" and r["config"]["frames_per_step_all_gpus"] == 109206, r
    assert r["config"]["frames_per_gpu"] < 109206 * 0.51, r
    assert r["redo_count"] == 0 and "k_fused_all" in r["config"]["kernel"], r
    assert r["value"] == pytest.approx(109206 * 2 / (r["ms_per_step"] * 2 / 1e3), rel=1e-6)


def test_non_finite_partials_cross_the_exchange_unchanged():
    """k_fused_all tags every published partial sum in its lowest mantissa bit; an infinite partial must
    come out of the exchange as the same infinity (readers clear the bit), so that results for overflowing inputs do
    not depend on which kernel ran.  One frame of X holds an inf: its activations go inf, then NaN - the same in the
    kernels that exchange and in those that do not; the other frames are untouched."""
    import exemplars_vc_amd as evc
    o = oracle()
    p = o.synth_problem(25, 4096, 688, seed=21)
    X = p["X"].copy()
    X[3, 100] = np.inf
    kw = dict(iters=6, eps_mode="zero_replace", init="const", init_value=0.01, info=True)
    a, ia = evc.solve_activations(p["A"], X, **kw)
    b, ib = evc.solve_activations(p["A"], X, cooperative=False, **kw)
    assert ia["kernel"] == "k_fused_all" and ia["exchange"] == 1 and ib["exchange"] == 0
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
    assert not np.isfinite(a[:, 100]).all() and np.isfinite(np.delete(a, 100, axis=1)).all()
    fin = np.isfinite(a)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-10)
