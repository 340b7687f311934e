"""STFT front end: evc_stft through the C ABI against the oracle's restatement of
librosa.core.stft as called at 04_align_n_nmf.py:422 (librosa itself is absent: parity with the package
is UNPINNED; the |Re STFT| of real audio in tests/golden/sklearn_audio_stft.npz was produced by the same
restated algorithm).  float64; tolerance 1e-11 of the largest magnitude (a 400-term DFT sum in another
order)."""
import numpy as np
import pytest

import exemplars_vc_amd as evc
from exemplars_vc_amd.compat import features
from oracle import evc_oracle as o

pytestmark = pytest.mark.gpu


def _signal(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    return 0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1330 * t + 1.0) + 0.05 * rng.standard_normal(n)


@pytest.mark.parametrize("n,n_fft,hop,center", [(55000, 400, 80, True), (4001, 400, 80, True), (401, 400, 80, False),
                                                (1000, 256, 64, True), (300, 400, 80, True), (2000, 30, 7, False),
                                                (201, 400, 80, True), (120, 400, 80, True), (16000, 512, 128, False)])
def test_stft_matches_restated_librosa(n, n_fft, hop, center):
    y = _signal(n, n)
    want = o.librosa_stft(y, n_fft, hop, center).T
    re, im = evc.stft(y, n_fft, hop, center=center)
    assert re.shape == want.shape == im.shape
    if want.size:
        scale = np.abs(want).max()
        assert np.abs(re - want.real).max() <= 1e-11 * scale
        assert np.abs(im - want.imag).max() <= 1e-11 * scale


def test_stft_shorter_than_a_frame_and_empty():
    re, im = evc.stft(np.ones(100), 400, 80, center=False)
    assert re.shape == (0, 201) and im.shape == (0, 201)
    re, im = evc.stft(np.zeros(0), 400, 80)
    assert re.shape == (0, 201)
    with pytest.raises(ValueError):
        evc.stft(np.ones(100), 401, 80)


def test_features_feed_the_solver_like_the_script():
    """extract_feature_for_conversion() -> factorize(): |real| of the complex64 STFT is float32, so the
    solve runs in float32, exactly as it would behind librosa (04_align_n_nmf.py:422-427,315-326)."""
    from exemplars_vc_amd.compat.factorize import factorize
    y = _signal(8000, 1)
    f = features.stft_features(y)
    assert f["stft"].dtype == np.complex64 and f["real"].dtype == np.float32 and f["stft"].shape == (101, 201)
    want = o.librosa_stft(y).T.astype(np.complex64)
    np.testing.assert_allclose(f["stft"], want, rtol=0, atol=2e-6 * np.abs(want).max())
    src = [{"real": features.stft_features(_signal(3000, s))["real"]} for s in (2, 3)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        H, R = factorize(f, src, use_stft=True)
    assert R is None and H["H_stft"].shape == (76, 101) and H["H_stft"].dtype == np.float32
    A = np.abs(np.concatenate([s["real"] for s in src]))
    act, _, _ = o.sklearn_mu_fixed_dictionary(np.abs(f["real"]).astype(np.float64), A.astype(np.float64), 150, 1e-4)
    # float32 solve vs float64 oracle on the same float32 inputs
    err = np.linalg.norm(H["H_stft"].T - act) / np.linalg.norm(act)
    assert err < 5e-3, err
    d64 = features.conversion_features(y, 16000, dtype=np.complex128)
    assert d64["fs"] == 16000 and d64["stft"].dtype == np.complex128
