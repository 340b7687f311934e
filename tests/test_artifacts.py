"""On-disk artefacts of the reference pipeline: compat.artifacts reads/writes the pickles of
04_align_n_nmf.py:65-85,251-260,296-308 / 03_a_b_r_parallel.py:122-153 / 01_make_dict_parallel.py:325-339
without executing anything from a file.  No pickle ships with the reference: the files read here are
written by this test with plain `pickle.dump(..., protocol=3)`, as the reference does."""
import os
import pickle

import numpy as np
import pytest

from exemplars_vc_amd.compat import artifacts as art
from exemplars_vc_amd.compat import factorize as fz
from oracle import evc_oracle as o


def _features(rng, n_files, use_stft):
    feats = []
    for _ in range(n_files):
        T = int(rng.integers(5, 9))
        if use_stft:
            feats.append({"stft": (rng.random((T, 201)) + 1j * rng.random((T, 201))).astype(np.complex64),
                          "fs": 16000})
        else:
            feats.append({"sp": rng.random((T, 513)), "ap": rng.random((T, 513)), "f0": rng.random(T) * 200,
                          "fs": 16000, "sr": 16000})
    return feats


@pytest.mark.parametrize("use_stft", [True, False])
def test_reference_layout_round_trip(tmp_path, use_stft):
    rng = np.random.default_rng(3)
    root = str(tmp_path)
    d = os.path.join(root, art.EXEM_DICT)
    os.makedirs(d)
    src, tar = _features(rng, 3, use_stft), _features(rng, 3, use_stft)
    W_A = [np.sort(rng.integers(0, 5, 7)).astype(np.int64) for _ in range(3)]
    W_B = [np.sort(rng.integers(0, 5, 7)).astype(np.int64) for _ in range(3)]
    # written exactly as the reference writes them
    suffix = "feat_stft.pkl" if use_stft else "feat_sp_ap_f0.pkl"
    for spk, feats in (("SF1", src), ("TF1", tar)):
        with open(os.path.join(d, f"{spk}_{suffix}"), "wb") as f:
            pickle.dump(feats, f, protocol=3)
    for name, W in (("exemplar_W_A", W_A), ("exemplar_W_B", W_B)):
        with open(os.path.join(d, name), "wb") as f:
            pickle.dump(W, f, protocol=3)
    s2, t2, wa, wb = art.io_load_from_pickle(root, "SF1", "TF1", use_stft)
    for got, want in ((s2, src), (t2, tar)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert set(g) == set(w)
            for k in w:
                np.testing.assert_array_equal(g[k], w[k])
                if isinstance(w[k], np.ndarray):
                    assert g[k].dtype == w[k].dtype
    for got, want in ((wa, W_A), (wb, W_B)):
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    # and what we write, plain pickle (the reference's reader) reads back identically
    art.write_features(root, "SM1", src, use_stft)
    art.write_exemplar_paths(root, W_A, W_B)
    with open(art.feature_path(root, "SM1", use_stft), "rb") as f:
        again = pickle.load(f)          # our own file
    key = "stft" if use_stft else "sp"
    np.testing.assert_array_equal(again[1][key], src[1][key])


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo should-never-run",))


def test_reader_executes_nothing(tmp_path):
    p = tmp_path / "bad.pkl"
    p.write_bytes(pickle.dumps([{"stft": _Evil(), "fs": 1}], protocol=3))
    with pytest.raises(art.UnsafeArtifactError):
        art.safe_load(str(p))
    p.write_bytes(pickle.dumps(np.array([{"a": 1}, None], dtype=object), protocol=3))
    with pytest.raises(art.UnsafeArtifactError):
        art.safe_load(str(p))
    p.write_bytes(pickle.dumps({"f": print}, protocol=3))
    with pytest.raises(art.UnsafeArtifactError):
        art.safe_load(str(p))
    # malformed structures are ValueErrors, not crashes
    d = tmp_path / art.EXEM_DICT
    os.makedirs(d)
    (d / "SF1_feat_stft.pkl").write_bytes(pickle.dumps({"not": "a list"}, protocol=3))
    with pytest.raises(ValueError):
        art.read_features(str(tmp_path), "SF1", True)
    (d / "exemplar_W_A").write_bytes(pickle.dumps([np.arange(3)], protocol=3))
    (d / "exemplar_W_B").write_bytes(pickle.dumps([np.arange(4)], protocol=3))
    with pytest.raises(ValueError):
        art.read_exemplar_paths(str(tmp_path))


def test_cache_names():
    assert art.activation_cache_path("r", True, 10).endswith(os.path.join("data", "vc", "exem_dict", "H_test_stft_10.pkl"))
    assert art.activation_cache_path("r", False, 162, "R").endswith("R_test_sp_ap_f0_162.pkl")
    assert art.activation_cache_path("r", False, 5, "H", "content", "abc").endswith("H_test_sp_ap_f0_5_abc.pkl")
    with pytest.raises(ValueError):
        art.activation_cache_path("r", True, 5, "R")
    a = np.arange(6.0).reshape(2, 3)
    assert art.content_digest(a) == art.content_digest(a.copy())
    assert art.content_digest(a) != art.content_digest(a.T)
    assert art.content_digest(a) != art.content_digest(a.astype(np.float32))


def _oracle_factorize(X, W, beta_loss="kullback-leibler", tol=1e-4, **kw):
    act, _, _ = o.sklearn_mu_fixed_dictionary(np.asarray(X), np.asarray(W), 150, tol)
    return act.T


@pytest.mark.parametrize("use_stft", [True, False])
def test_factorize_cache_semantics(tmp_path, monkeypatch, use_stft):
    """Host logic of the cache around the solve (the solve itself is the oracle here: no GPU)."""
    calls = []

    def counting(X, W, **kw):
        calls.append(X.shape)
        return _oracle_factorize(X, W, **kw)

    monkeypatch.setattr(fz, "_factorize", counting)
    monkeypatch.setattr(fz, "synthesize_rows", lambda H, B, device=None: o.s4_convert(H, np.asarray(B)))

    def counting_recon(conv, A, tol, device, hint):      # the WORLD branch's seam: H and H.T @ A from one call
        H = counting(conv, A, tol=tol)
        return H, o.s4_convert(H, np.asarray(A))

    monkeypatch.setattr(fz, "_factorize_recon", counting_recon)
    rng = np.random.default_rng(5)
    if use_stft:
        src = [{"real": rng.random((6, 9)) - 0.3} for _ in range(2)]
        utt1, utt2 = {"real": rng.random((4, 9))}, {"real": rng.random((4, 9))}
    else:
        src = [{"sp": rng.random((6, 9)) + 0.1, "ap": rng.random((6, 9)) + 0.1, "f0": rng.random(6) + 0.1}
               for _ in range(2)]
        utt1, utt2 = ({"sp": rng.random((4, 9)), "ap": rng.random((4, 9)), "f0": rng.random(4)} for _ in range(2))
    per_call = 1 if use_stft else 3
    root = str(tmp_path)
    H1, R1 = fz.factorize(utt1, src, use_stft=use_stft, cache_dir=root)
    assert len(calls) == per_call
    H1b, R1b = fz.factorize(utt1, src, use_stft=use_stft, cache_dir=root)           # served from the file
    assert len(calls) == per_call
    for k in H1:
        np.testing.assert_array_equal(H1[k], H1b[k])
    if not use_stft:
        for k in R1:
            np.testing.assert_array_equal(R1[k], R1b[k])
    H2, _ = fz.factorize(utt2, src, use_stft=use_stft, cache_dir=root)              # another utterance: solved
    assert len(calls) == 2 * per_call
    k0 = next(iter(H1))
    assert not np.array_equal(H1[k0], H2[k0])
    # the reference's key: same name for every utterance -> the stale H comes back (reproduced on request)
    Hr1, _ = fz.factorize(utt1, src, use_stft=use_stft, cache_dir=root, cache_key="reference")
    Hr2, _ = fz.factorize(utt2, src, use_stft=use_stft, cache_dir=root, cache_key="reference")
    np.testing.assert_array_equal(Hr1[k0], Hr2[k0])
    assert os.path.isfile(art.activation_cache_path(root, use_stft, 2))
    # what the reference wrote is what its own reader (plain pickle, our file) returns
    with open(art.activation_cache_path(root, use_stft, 2), "rb") as f:
        raw = pickle.load(f)
    np.testing.assert_array_equal(raw[k0], Hr1[k0])
    if not use_stft:
        # H on disk but R missing: R is recomputed from the cached H, no new solve
        os.remove(art.activation_cache_path(root, False, 2, "R"))
        n = len(calls)
        _, Rr = fz.factorize(utt1, src, use_stft=False, cache_dir=root, cache_key="reference")
        assert len(calls) == n and set(Rr) == {"r_sp", "r_ap", "r_f0"}
        assert os.path.isfile(art.activation_cache_path(root, False, 2, "R"))


def test_log_ratio_residual_extension(monkeypatch):
    """factorize(residual='log_ratio') / convert(residual_mode='log_ratio') (not in the
    reference): y_hat_target * y / y_hat_source on floored positives.  Host logic; the solve is the oracle."""
    monkeypatch.setattr(fz, "_factorize", _oracle_factorize)
    monkeypatch.setattr(fz, "synthesize_rows", lambda H, B, device=None: o.s4_convert(H, np.asarray(B)))
    monkeypatch.setattr(fz, "_factorize_recon", lambda conv, A, tol, device, hint: (
        _oracle_factorize(conv, A, tol=tol), o.s4_convert(_oracle_factorize(conv, A, tol=tol), np.asarray(A))))
    rng = np.random.default_rng(11)
    src = [{"sp": rng.random((6, 9)) + 0.1, "ap": rng.random((6, 9)) + 0.1, "f0": rng.random(6) + 0.1} for _ in range(2)]
    tar = [{"sp": rng.random((6, 9)) + 0.1, "ap": rng.random((6, 9)) + 0.1, "f0": rng.random(6) + 0.1} for _ in range(2)]
    utt = {"sp": rng.random((4, 9)) + 0.05, "ap": rng.random((4, 9)) + 0.05, "f0": rng.random(4) + 0.05}
    H, R = fz.factorize(utt, src, use_stft=False, residual="log_ratio")
    assert all(np.isfinite(R[k]).all() for k in R)
    out = fz.convert(H, tar, R, use_stft=False, residual_mode="log_ratio")
    for name in ("sp", "ap"):
        A = np.concatenate([f[name] for f in src]); B = np.concatenate([f[name] for f in tar])
        want = (H["H_" + name].T @ B) * utt[name] / (H["H_" + name].T @ A)
        np.testing.assert_allclose(out[name], want, rtol=1e-12)
    assert out["f0"].shape == (4,)
    # converting with the source's own exemplars returns the source exactly: the residual closes the gap
    back = fz.convert(H, src, R, use_stft=False, residual_mode="log_ratio")
    np.testing.assert_allclose(back["sp"], utt["sp"], rtol=1e-12)
    with pytest.raises(ValueError):
        fz.factorize(utt, src, use_stft=False, residual="nope")
