"""Griffin-Lim on the GPU against vectors produced by the reference's own
zz_audio_utilities.reconstruct_signal_griffin_lim, and against the oracle at the size the
scripts use (688 frames, fft 400, hop 80).  `-m gpu`.

Tolerance: float64 throughout; the GPU forms the transforms as dense DFT contractions where
numpy uses pocketfft, so single transforms agree to ~1e-15 and the iteration (a non-expansive
projection) keeps the difference at ~1e-12 after hundreds of iterations.  Asserted:
max |x_gpu - x_ref| <= 1e-9 * max |x_ref|.
"""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def close(got, want, tol=1e-9):
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= tol, err


@pytest.mark.parametrize("path", golden_files("gl_"), ids=os.path.basename)
def test_reference_vectors(path):
    import exemplars_vc_amd as evc
    g = load_golden(path)
    x, rmse = evc.griffin_lim(g["mag"], int(g["n_fft"]), int(g["hop"]), int(g["iters"]), g["x0"], want_rmse=True)
    close(x, g["x"])
    np.testing.assert_allclose(rmse, g["rmse"], rtol=1e-8)


def test_drop_in_surface_uses_the_global_rng_like_the_reference(capsys):
    from exemplars_vc_amd.compat.griffin_lim import reconstruct_signal_griffin_lim
    g = load_golden([p for p in golden_files("gl_") if "k25" in p][0])
    np.random.seed(11)                                  # the seed tools/make_golden.py used for this case
    x = reconstruct_signal_griffin_lim(g["mag"], int(g["n_fft"]), int(g["hop"]), int(g["iters"]))
    close(x, g["x"])
    out = capsys.readouterr().out
    assert out.count("Reconstruction iteration") == int(g["iters"])


def test_full_utterance_size_and_zero_magnitudes():
    import exemplars_vc_amd as evc
    from oracle import evc_oracle as o
    rng = np.random.default_rng(3)
    T, F, hop, K = 688, 400, 80, 12
    mag = rng.random((T, F // 2 + 1)) ** 3
    mag[100:110] = 0.0                                  # silent frames: S = 0 -> angle 0
    x0 = rng.standard_normal(T * hop + F)
    want, tr = o.griffin_lim(mag, F, hop, K, x0)
    got, rm = evc.griffin_lim(mag, F, hop, K, x0, want_rmse=True)
    close(got, want)
    np.testing.assert_allclose(rm, tr, rtol=1e-8)
    assert np.all(np.diff(rm[1:]) <= 0)                 # Griffin-Lim's consistency error is non-increasing
    assert np.array_equal(evc.griffin_lim(mag, F, hop, 0, x0), x0)
    with pytest.raises(ValueError):
        evc.griffin_lim(mag[:, :-1], F, hop, 1, x0)


def test_batch_of_ragged_utterances_equals_single_calls():
    """evc_griffin_lim_batch: utterances of different lengths (one shorter than a frame hop group, one silent)
    laid out as one virtual signal; every result against the single-utterance call and against the oracle."""
    import exemplars_vc_amd as evc
    from oracle import evc_oracle as o
    rng = np.random.default_rng(17)
    F, hop, K = 400, 80, 10
    Ts = [688, 1, 216, 3, 945, 64]
    mags = [rng.random((T, F // 2 + 1)) ** 2 for T in Ts]
    mags[3][:] = 0.0
    x0s = [rng.standard_normal(T * hop + F) for T in Ts]
    outs, rmse = evc.griffin_lim_batch(mags, F, hop, K, x0s, want_rmse=True)
    assert rmse.shape == (len(Ts), K)
    for u, T in enumerate(Ts):
        single, rm = evc.griffin_lim(mags[u], F, hop, K, x0s[u], want_rmse=True)
        scale = max(np.max(np.abs(single)), 1e-300)
        assert np.max(np.abs(outs[u] - single)) <= 1e-11 * scale, u
        np.testing.assert_allclose(rmse[u], rm, rtol=1e-9, atol=1e-300)
    want, _ = o.griffin_lim(mags[2], F, hop, K, x0s[2])
    close(outs[2], want)
    # zero iterations hand the initial signals back; a batch of one is the single call, bit for bit
    same = evc.griffin_lim_batch(mags, F, hop, 0, x0s)
    assert all(np.array_equal(a, b) for a, b in zip(same, x0s))
    one = evc.griffin_lim_batch(mags[:1], F, hop, K, x0s[:1])[0]
    assert np.array_equal(one, evc.griffin_lim(mags[0], F, hop, K, x0s[0]))
    with pytest.raises(ValueError):
        evc.griffin_lim_batch(mags, F, hop, 1, x0s[:-1])
