#!/usr/bin/env python3
"""The reference's STFT pipeline end to end on the GPU, on synthetic "voices" (no audio ships with this repo).

  01_make_dict_parallel.py   DTW paths of the parallel training pairs          compat.make_dict
  03_a_b_r_parallel.py       STFT features of every training utterance        compat.features
  (pickles in data/vc/exem_dict/)                                            compat.artifacts
  04_align_n_nmf.py          gather aligned exemplars, H = _factorize(X, A),  compat.make_dict / compat.factorize
                             Y = H.T @ B, Griffin-Lim                         compat.griffin_lim

The DTW features of the reference are librosa MFCCs (absent here); a 25-band log-magnitude stands in.
usage: python examples/pipeline_synthetic.py [n_pairs] [seconds per utterance]
"""
import os
import sys
import tempfile
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exemplars_vc_amd.compat import artifacts, factorize as fz, features, griffin_lim, make_dict  # noqa: E402

FS = 16000


def voice(seconds, f0, formants, warp, seed):
    """A harmonic source with a slowly moving pitch through a few resonances; `warp` stretches time."""
    rng = np.random.default_rng(seed)
    n = int(seconds * FS * warp)
    t = np.arange(n) / FS / warp
    pitch = f0 * (1.0 + 0.15 * np.sin(2 * np.pi * 0.7 * t) + 0.05 * np.sin(2 * np.pi * 2.3 * t + seed))
    phase = 2 * np.pi * np.cumsum(pitch) / FS
    y = np.zeros(n)
    for h in range(1, 30):
        fh = h * f0
        gain = sum(np.exp(-0.5 * ((fh - fc) / bw) ** 2) for fc, bw in formants) + 0.02
        y += gain / h ** 0.5 * np.sin(h * phase)
    env = 0.6 + 0.4 * np.sin(2 * np.pi * 1.3 * t) ** 2
    return 0.2 * env * y + 0.002 * rng.standard_normal(n)


def band_log_mag(stft_rows, bands=25):
    mag = np.abs(stft_rows)
    edges = np.linspace(0, mag.shape[1], bands + 1).astype(int)
    return np.log(np.stack([mag[:, a:b].mean(axis=1) for a, b in zip(edges[:-1], edges[1:])], axis=1) + 1e-6)


def spectral_distance(a, b):
    n = min(len(a), len(b))
    return float(np.mean(np.abs(np.log(np.abs(a[:n]) + 1e-4) - np.log(np.abs(b[:n]) + 1e-4))))


def main(n_pairs=6, seconds=1.0, gl_iters=60, verbose=True):
    src_f = [(700, 130), (1200, 160), (2600, 250)]
    tar_f = [(850, 130), (1500, 180), (2900, 250)]
    say = print if verbose else (lambda *a, **k: None)
    t0 = time.perf_counter()
    src_wavs = [voice(seconds, 120 + 6 * i, src_f, 1.0, 10 + i) for i in range(n_pairs)]
    tar_wavs = [voice(seconds, 210 + 9 * i, tar_f, 1.1 + 0.02 * i, 50 + i) for i in range(n_pairs)]

    # 03_a_b_r_parallel.py: features of the training set
    src_feat = [features.conversion_features(w, FS) for w in src_wavs]
    tar_feat = [features.conversion_features(w, FS) for w in tar_wavs]
    # 01_make_dict_parallel.py: DTW paths (features transposed to (order, frames) as the reference holds them)
    paths, _, _ = make_dict.dtw_alignment([band_log_mag(f["stft"]).T for f in src_feat],
                                          [band_log_mag(f["stft"]).T for f in tar_feat])
    W_A, W_B = make_dict.make_exemplar_dict_W(paths)
    t_dict = time.perf_counter() - t0

    with tempfile.TemporaryDirectory() as root:
        # the artefacts the reference's scripts exchange on disk
        artifacts.write_features(root, "SF1", src_feat, True)
        artifacts.write_features(root, "TF1", tar_feat, True)
        artifacts.write_exemplar_paths(root, W_A, W_B)
        s_feat, t_feat, s_W, t_W = artifacts.io_load_from_pickle(root, "SF1", "TF1", True)
        aligned_src, aligned_tar = make_dict.align_sp_ap_f0(s_feat, t_feat, s_W, t_W, use_stft=True)
        N = sum(len(f["real"]) for f in aligned_src)

        # 04_align_n_nmf.py __main__: convert one unseen utterance
        utt = voice(seconds, 125, src_f, 1.0, 999)
        truth = voice(seconds, 215, tar_f, 1.0, 998)
        t1 = time.perf_counter()
        tobe = features.stft_features(utt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # ConvergenceWarning at 150 iterations, as in the reference
            H, R = fz.factorize(tobe, aligned_src, use_stft=True, cache_dir=root)
        converted = fz.convert(H, aligned_tar, residual=R, use_stft=True)
        t_conv = time.perf_counter() - t1
        # the same dictionary built in one pass on the GPU (DTW paths consumed on the device, aligned frames gathered
        # there, images prepared once): what a caller converting many utterances would hold
        from exemplars_vc_amd import convert as evc_convert
        t3 = time.perf_counter()
        pd, rows = make_dict.aligned_dictionary([band_log_mag(f["stft"]).T for f in src_feat],
                                                [band_log_mag(f["stft"]).T for f in tar_feat], src_feat, tar_feat)
        t_pd = time.perf_counter() - t3
        assert int(rows[-1]) == N
        H_d, Y_d = evc_convert(pd, np.abs(np.asarray(tobe["real"])).astype(H["H_stft"].dtype), layout="frame_major",
                               iters=150, eps_mode="zero_replace", init="sklearn", check_every=10, stop_rule="sklearn",
                               tol=1e-4)
        assert np.allclose(Y_d, converted, rtol=1e-3, atol=1e-6 * float(np.abs(converted).max()))
        say(f"dictionary on the device (DTW + gather + prepare, no frame leaves the GPU): {t_pd * 1e3:.1f} ms")
        np.random.seed(0)
        t2 = time.perf_counter()
        wav, wav_path = griffin_lim.synthesize2(converted, FS, "converted", out_dir=os.path.join(root, "wav"),
                                                iterations=gl_iters)
        t_gl = time.perf_counter() - t2
        from scipy.io import wavfile
        sr_back, wav_back = wavfile.read(wav_path)
        assert sr_back == FS and wav_back.dtype == np.float32 and len(wav_back) == len(wav)

    want = np.abs(features.stft_features(truth)["real"])
    d_before = spectral_distance(np.abs(tobe["real"]), want)
    d_after = spectral_distance(converted, want)
    say(f"dictionary: {n_pairs} pairs, {N} exemplars (DTW + features {t_dict * 1e3:.1f} ms)")
    say(f"conversion: {converted.shape[0]} frames x {converted.shape[1]} bins, H {H['H_stft'].shape} "
        f"{H['H_stft'].dtype} ({t_conv * 1e3:.1f} ms); Griffin-Lim {gl_iters} iterations ({t_gl * 1e3:.1f} ms)")
    say(f"log-spectral distance to the target speaker: source {d_before:.3f} -> converted {d_after:.3f}")
    return {"N": N, "converted": converted, "wav": wav, "H": H["H_stft"], "d_before": d_before, "d_after": d_after}


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 6, float(a[1]) if len(a) > 1 else 1.0)
