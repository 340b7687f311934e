"""Griffin-Lim back end of the reference, on the GPU.

Mirrors /root/reference/zz_audio_utilities.py:258-292 `reconstruct_signal_griffin_lim`
(called by synthesize2(), 04_align_n_nmf.py:182-191): same arguments, the initial signal is
drawn with np.random.randn from the global numpy RNG exactly as the reference does, and the
per-iteration RMSE lines are printed (after the run, since the iterations execute on the device).
"""
from __future__ import annotations

import numpy as np

from ..solver import griffin_lim


def reconstruct_signal_griffin_lim(magnitude_spectrogram, fft_size, hopsamp, iterations, *, device=None,
                                   verbose=True):
    magnitude_spectrogram = np.asarray(magnitude_spectrogram, dtype=np.float64)
    time_slices = magnitude_spectrogram.shape[0]
    len_samples = int(time_slices * hopsamp + fft_size)
    x_reconstruct = np.random.randn(len_samples)          # zz_audio_utilities.py:279
    x, rmse = griffin_lim(magnitude_spectrogram, fft_size, hopsamp, iterations, x_reconstruct,
                          device=device, want_rmse=True)
    if verbose:
        for i, diff in enumerate(rmse):
            print('Reconstruction iteration: {}/{} RMSE: {} '.format(i + 1, iterations, diff))
    return x


def synthesize2(stft_mag, fs, filename=None, *, out_dir="wav", iterations=300, frame_length=400, hop_length=80,
                device=None, verbose=False):
    """`synthesize2(stft_mag, fs, filename)` of 04_align_n_nmf.py:181-191: Griffin-Lim on |stft_mag| (300
    iterations, frame 400, hop 80) and the result written to `wav/<filename>.wav`.  librosa.output.write_wav
    (absent here) stored the float signal as a 32-bit float wav without normalising; scipy.io.wavfile does the
    same.  Returns (y, path); filename=None skips the file."""
    y = reconstruct_signal_griffin_lim(np.abs(np.asarray(stft_mag, dtype=np.float64)), frame_length, hop_length,
                                       iterations, device=device, verbose=verbose)
    path = None
    if filename is not None:
        import os
        from scipy.io import wavfile
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"{filename}.wav")
        wavfile.write(path, int(fs), y.astype(np.float32))
    return y, path
