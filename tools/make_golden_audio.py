#!/usr/bin/env python3
"""Generate tests/golden/audio_stft_n4096_{f32,f64}.npz - the script's default flow at the size it runs it
(SURVEY.md 8d "real-audio variant").  Runs ONLY in the build container (it reads the
reference's sample audio and calls the installed scikit-learn; neither travels to the GPU box).

  inputs   |Re STFT| (n_fft 400, hop 80, periodic hann, reflect-centred: the librosa.stft call of
           04_align_n_nmf.py:422, restated in numpy) of data/{SF1,TF1}/10000{1..8}.wav; each pair aligned by the
           restated `dtw` package algorithm (oracle.dtw_align) on log-magnitude frames; the aligned frame pairs of
           all eight files in file order, the first 4096 of them: A = source rows, B = target rows
           (04_align_n_nmf.py:100-169,320-324,354-361); X = |Re STFT| of wav/SF1_100162.wav (T = 688).
           Stored as float32 - the script's STFT is complex64 - and used as they are by both runs.
  outputs  the installed scikit-learn through exactly the call of 04_align_n_nmf.py:212-213 (tol 1e-4, <= 150
           iterations) in float32 and in float64 (the float32 values widened): n_iter, Y = H.T @ B (T x 201) and the
           first 32 frames of H (N x 32).

Only arrays and scalars are written: derived magnitudes, alignment-selected rows and results - no source text.
"""
import os
import sys
import time
import wave
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
N_EXEMPLARS = 4096


def read_wav(path):
    with wave.open(path) as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1 and w.getframerate() == 16000
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return pcm.astype(np.float64) / 32768.0


def stft_rows(y, n_fft=400, hop=80):
    """complex STFT, frames as rows (librosa.stft(...).T with librosa's defaults)"""
    y = np.pad(y, n_fft // 2, mode="reflect")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(y[idx] * win, axis=1)


def run_sklearn(X_rows, W_rows, tol=1e-4, max_iter=150):
    from sklearn.decomposition import non_negative_factorization
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _W, _H, n_iter = non_negative_factorization(
            X=X_rows, H=W_rows, init="custom", update_H=False, n_components=W_rows.shape[0],
            beta_loss="frobenius", solver="mu", tol=tol, max_iter=max_iter, verbose=0)
    return _W.T, int(n_iter)          # what _factorize returns: N x T


def main():
    from oracle import evc_oracle as o
    a_rows, b_rows, path_len = [], [], []
    for i in range(1, 9):
        S = stft_rows(read_wav(os.path.join(REF, f"data/SF1/10000{i}.wav")))
        Tg = stft_rows(read_wav(os.path.join(REF, f"data/TF1/10000{i}.wav")))
        fa, fb = np.log(np.abs(S) + 1e-8), np.log(np.abs(Tg) + 1e-8)
        _, (pa, pb) = o.dtw_align(fa, fb)
        a_rows.append(np.abs(S.real)[pa])
        b_rows.append(np.abs(Tg.real)[pb])
        path_len.append(len(pa))
        print(f"pair {i}: {S.shape[0]} x {Tg.shape[0]} frames -> {len(pa)} aligned pairs", flush=True)
    A = np.concatenate(a_rows)[:N_EXEMPLARS].astype(np.float32)
    B = np.concatenate(b_rows)[:N_EXEMPLARS].astype(np.float32)
    X = np.abs(stft_rows(read_wav(os.path.join(REF, "wav/SF1_100162.wav"))).real).astype(np.float32)
    assert A.shape == (N_EXEMPLARS, 201) and X.shape[1] == 201, (A.shape, X.shape)
    print("A", A.shape, "X", X.shape, "aligned pairs per file", path_len, flush=True)
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        t0 = time.time()
        H, n_iter = run_sklearn(X.astype(dt), A.astype(dt))
        Y = H.T @ B.astype(dt)
        print(f"{tag}: n_iter {n_iter}, {time.time() - t0:.1f} s, H {H.dtype}", flush=True)
        extra = dict(A_rows=A, B_rows=B, X_rows=X) if tag == "f32" else {}     # the inputs are stored once
        np.savez_compressed(os.path.join(OUT, f"audio_stft_n4096_{tag}.npz"), n_iter=n_iter, tol=1e-4, max_iter=150,
                            Y_rows=Y, H_first32=np.ascontiguousarray(H[:, :32]),
                            err_final=float(np.linalg.norm(X.astype(dt) - H.T @ A.astype(dt))), **extra)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("audio_stft_n4096"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
