"""STFT feature extraction of the conversion script, on the GPU (front half).

Mirrors the STFT branch of extract_feature_for_conversion() (04_align_n_nmf.py:419-429) and of
_get_conversion_data() (03_a_b_r_parallel.py:101-104) from the decoded samples on: reading the wav file
(librosa.load) stays with the caller.  librosa is absent in this environment, so the window / padding
conventions follow its published algorithm (restated for the tests as `librosa_stft`): parity with librosa
itself is unpinned.
"""
from __future__ import annotations

import numpy as np

from ..solver import stft

FRAME_LENGTH = 400      # 04_align_n_nmf.py:46
HOP_LENGTH = 80         # 04_align_n_nmf.py:47


def stft_features(samples, *, n_fft=FRAME_LENGTH, hop_length=HOP_LENGTH, dtype=np.complex64, device=None):
    """-> {'stft': (T, 201) complex, 'real': ..., 'imag': ...}: `feat_stft.T` and its parts, as
    extract_feature_for_conversion() returns them.  dtype=np.complex64 is librosa's default output type
    (so 'real' is float32 and the solve downstream runs in float32, as it does in the reference);
    dtype=np.complex128 keeps the float64 the transform is computed in."""
    re, im = stft(np.asarray(samples, dtype=np.float64), n_fft, hop_length, center=True, device=device)
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
        raise ValueError("dtype must be complex64 or complex128")
    z = (re + 1j * im).astype(dtype)
    return {"stft": z, "real": np.real(z), "imag": np.imag(z)}


def conversion_features(samples, fs, *, n_fft=FRAME_LENGTH, hop_length=HOP_LENGTH, dtype=np.complex64, device=None):
    """The per-file dict of `<spk>_feat_stft.pkl`: {'stft': (T, 201) complex, 'fs': fs}
    (03_a_b_r_parallel.py:101-104)."""
    return {"stft": stft_features(samples, n_fft=n_fft, hop_length=hop_length, dtype=dtype, device=device)["stft"],
            "fs": fs}
