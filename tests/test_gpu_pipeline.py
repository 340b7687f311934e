"""All rows of the scope table chained the way the reference's scripts chain them (STFT features -> DTW
dictionary -> on-disk artefacts -> factorize -> convert -> Griffin-Lim), on synthetic voices.  Every stage
has its own parity test; this one checks that they compose: shapes, dtypes, the float32 flow the
complex64 STFT implies, and that the conversion moves the spectrum towards the target speaker."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_pipeline_composes():
    import pipeline_synthetic as ps
    out = ps.main(n_pairs=4, seconds=0.5, gl_iters=10, verbose=False)
    T = 1 + int(0.5 * ps.FS) // 80
    assert out["converted"].shape == (T, 201) and out["H"].shape == (out["N"], T)
    assert out["H"].dtype == np.float32            # |real| of a complex64 STFT, as behind librosa
    assert np.isfinite(out["converted"]).all() and (out["converted"] >= 0).all()
    assert out["wav"].shape == (T * 80 + 400,) and np.isfinite(out["wav"]).all()
    assert out["d_after"] < out["d_before"], (out["d_before"], out["d_after"])
