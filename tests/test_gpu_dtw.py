"""DTW alignment on the GPU against the restated `dtw` package algorithm.  `-m gpu`.

PARITY UNPINNED with respect to the package itself: it is not installable here and the reference holds
no alignment output.  What is asserted: index paths and accumulated costs are BIT-EXACT against
oracle.dtw_align (the published algorithm: costs sum(np.square(x - y)) summed left to right, steps
(i-1,j-1),(i-1,j),(i,j-1), trace-back ties to the diagonal then to i-1), plus the properties any DTW path
must have and an independently written dynamic programme for the optimal cost.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle():
    from oracle import evc_oracle
    return evc_oracle


def make_pairs(rng, shapes, D):
    A, B = [], []
    for ta, tb in shapes:
        base = np.cumsum(rng.standard_normal((max(ta, tb) + 8, D)), axis=0)
        ia = np.sort(rng.choice(len(base), ta, replace=True))
        ib = np.sort(rng.choice(len(base), tb, replace=True))
        A.append(base[ia] + 0.05 * rng.standard_normal((ta, D)))
        B.append(base[ib] + 0.05 * rng.standard_normal((tb, D)))
    return A, B


def test_paths_are_bit_exact_against_the_restated_algorithm():
    import exemplars_vc_amd as evc
    o = oracle()
    rng = np.random.default_rng(5)
    shapes = [(40, 25), (1, 1), (1, 17), (23, 1), (64, 64), (37, 90), (130, 97)]
    A, B = make_pairs(rng, shapes, 25)
    paths, cost = evc.dtw_align(A, B, want_cost=True)
    for (pa, pb), c, a, b in zip(paths, cost, A, B):
        D1, (qa, qb) = o.dtw_align(a, b)
        assert np.array_equal(pa, qa) and np.array_equal(pb, qb)
        assert c == D1[-1, -1]
        # properties of a warping path
        assert pa[0] == 0 and pb[0] == 0 and pa[-1] == len(a) - 1 and pb[-1] == len(b) - 1
        da, db = np.diff(pa), np.diff(pb)
        assert np.all((da >= 0) & (da <= 1) & (db >= 0) & (db <= 1) & (da + db >= 1))


def test_ties_follow_the_package_rule_and_cost_is_optimal():
    import exemplars_vc_amd as evc
    o = oracle()
    # integer features: many exact ties in the accumulated cost
    rng = np.random.default_rng(9)
    a = rng.integers(0, 3, (45, 4)).astype(np.float64)
    b = rng.integers(0, 3, (38, 4)).astype(np.float64)
    (pa, pb), = evc.dtw_align([a], [b])
    D1, (qa, qb) = o.dtw_align(a, b)
    assert np.array_equal(pa, qa) and np.array_equal(pb, qb)
    # independent DP for the optimal accumulated cost
    C = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    acc = np.full((len(a) + 1, len(b) + 1), np.inf)
    acc[0, 0] = 0.0
    for i in range(len(a)):
        for j in range(len(b)):
            acc[i + 1, j + 1] = C[i, j] + min(acc[i, j], acc[i, j + 1], acc[i + 1, j])
    assert C[pa, pb].sum() == acc[-1, -1]


def test_drop_in_surface_and_gather():
    from exemplars_vc_amd.compat.make_dict import _dtw_alignment, dtw_alignment, make_exemplar_dict_W, align_sp_ap_f0
    o = oracle()
    rng = np.random.default_rng(2)
    A, B = make_pairs(rng, [(50, 44), (31, 36)], 24)
    fa, fb = [x.T for x in A], [x.T for x in B]          # the reference keeps (order, n_frames)
    p0 = _dtw_alignment(fa[0], fb[0])
    paths, _, _ = dtw_alignment(fa, fb)
    _, (qa, qb) = o.dtw_align(A[0], B[0])
    assert np.array_equal(p0[0], qa) and np.array_equal(paths[0][1], qb)
    src_W, tar_W = make_exemplar_dict_W(paths)
    src_feat = [{"stft": rng.standard_normal((len(a), 9)) + 1j * rng.standard_normal((len(a), 9)), "fs": 16000} for a in A]
    tar_feat = [{"stft": rng.standard_normal((len(b), 9)) + 1j * rng.standard_normal((len(b), 9)), "fs": 16000} for b in B]
    s, t = align_sp_ap_f0(src_feat, tar_feat, src_W, tar_W, use_stft=True)
    assert len(s) == 2 and s[0]["real"].shape == (len(src_W[0]), 9) and t[1]["imag"].shape == (len(tar_W[1]), 9)
    assert np.array_equal(s[1]["stft"][3], src_feat[1]["stft"][src_W[1][3]])


def test_corpus_sized_batch():
    """162 pairs of ~700 x ~700 frames (the reference's corpus): every workgroup finishes, paths are valid and
    a sample of them is compared with the oracle."""
    import exemplars_vc_amd as evc
    o = oracle()
    rng = np.random.default_rng(1)
    shapes = [(int(rng.integers(200, 760)), int(rng.integers(200, 760))) for _ in range(162)]
    A, B = make_pairs(rng, shapes, 25)
    paths = evc.dtw_align(A, B)
    for k, ((pa, pb), a, b) in enumerate(zip(paths, A, B)):
        assert pa[0] == 0 and pb[0] == 0 and pa[-1] == len(a) - 1 and pb[-1] == len(b) - 1
        if k in (0, 80):
            _, (qa, qb) = o.dtw_align(a, b)
            assert np.array_equal(pa, qa) and np.array_equal(pb, qb)


def test_dictionary_built_on_the_device_equals_the_host_gather():
    """evc_dtw_path_rows + evc_dtw_gather_rows (compat.make_dict.aligned_dictionary): the aligned frames gathered on the
    device by the paths evc_dtw_align left there - bit for bit what align_sp_ap_f0 + the stacking of factorize() /
    convert() produce through host lists (04_align_n_nmf.py:100-169,230-246,320-324), for the STFT flow (|real| of complex
    frames) and for a WORLD stream; a solve from the prepared dictionary equals the solve from the host-built arrays."""
    import exemplars_vc_amd as evc
    from exemplars_vc_amd.compat import make_dict
    rng = np.random.default_rng(11)
    n_pairs, order = 7, 13
    la = [int(v) for v in rng.integers(30, 140, n_pairs)]
    lb = [int(v) for v in rng.integers(30, 140, n_pairs)]
    dtw_a = [rng.standard_normal((order, n)) for n in la]          # (order, frames) as the reference holds them
    dtw_b = [rng.standard_normal((order, n)) for n in lb]
    src = [{"stft": (rng.standard_normal((n, 201)) + 1j * rng.standard_normal((n, 201))), "sp": rng.random((n, 513)) + 0.1}
           for n in la]
    tar = [{"stft": (rng.standard_normal((n, 201)) + 1j * rng.standard_normal((n, 201))), "sp": rng.random((n, 513)) + 0.1}
           for n in lb]
    # the reference's step-by-step route (host lists)
    paths, _, _ = make_dict.dtw_alignment(dtw_a, dtw_b)
    W_A, W_B = make_dict.make_exemplar_dict_W(paths)
    want_A = np.concatenate([np.abs(np.asarray(f["stft"])[ia].real) for f, ia in zip(src, W_A)], axis=0)
    want_B = np.concatenate([np.abs(np.asarray(f["stft"])[ib].real) for f, ib in zip(tar, W_B)], axis=0)
    A, B, rows = evc.dtw_dictionary([a.T for a in dtw_a], [b.T for b in dtw_b], [f["stft"] for f in src],
                                    [f["stft"] for f in tar], op="abs", real_part=True)
    assert rows[-1] == len(want_A) and list(np.diff(rows)) == [len(p[0]) for p in paths]
    assert np.array_equal(A.cpu().numpy(), want_A) and np.array_equal(B.cpu().numpy(), want_B)
    # a WORLD stream (real frames, plain copy)
    A2, B2, _ = evc.dtw_dictionary([a.T for a in dtw_a], [b.T for b in dtw_b], [f["sp"] for f in src], [f["sp"] for f in tar])
    assert np.array_equal(A2.cpu().numpy(), np.concatenate([f["sp"][ia] for f, ia in zip(src, W_A)], axis=0))
    assert np.array_equal(B2.cpu().numpy(), np.concatenate([f["sp"][ib] for f, ib in zip(tar, W_B)], axis=0))
    # the prepared dictionary drives the solver like the host-built arrays do
    pd, rows2 = make_dict.aligned_dictionary(dtw_a, dtw_b, src, tar, use_stft=True)
    assert np.array_equal(rows, rows2)
    X = np.abs(rng.standard_normal((50, 201)))
    kw = dict(layout="frame_major", iters=12, eps_mode="zero_replace", init="sklearn")
    H1, Y1 = evc.convert(pd, X, **kw)
    H2, Y2 = evc.convert(want_A, X, want_B, **kw)
    assert np.array_equal(H1, H2) and np.array_equal(Y1, Y2)
