"""Dictionary construction of the reference on the GPU: DTW alignment and the gather of aligned frames.

Mirrors /root/reference/01_make_dict_parallel.py:215-249 (`_dtw_alignment`, `dtw_alignment`), :291-292
(`make_exemplar_dict_W`) and /root/reference/04_align_n_nmf.py:100-169 (`align_sp_ap_f0`).  The reference
calls the third-party `dtw` package; its algorithm is restated (see csrc/evc_dtw.hip) - parity with the
package itself is unpinned because it is not installable here.
"""
from __future__ import annotations

import numpy as np

from ..solver import dtw_align


def _dtw_alignment(feat_A, feat_B, *, device=None):
    """DTW path of one utterance pair.  feat_A, feat_B: (order, n_frames) as in the reference, which
    passes the transposes to dtw().  Returns (path_a, path_b)."""
    return dtw_align([np.asarray(feat_A).T], [np.asarray(feat_B).T], device=device)[0]


def dtw_alignment(feat_full_A, feat_full_B, *, device=None):
    """All pairs at once (the reference maps `_dtw_alignment` over a process pool): one workgroup per
    pair.  Returns (dtw_paths, None, None) like the reference."""
    paths = dtw_align([np.asarray(a).T for a in feat_full_A], [np.asarray(b).T for b in feat_full_B], device=device)
    return paths, None, None


def make_exemplar_dict_W(dtw_paths):
    return [path[0] for path in dtw_paths], [path[1] for path in dtw_paths]


def align_sp_ap_f0(src_feat, tar_feat, src_W, tar_W, *, use_stft=True):
    """Gather the aligned frames of every file by its DTW index lists (04_align_n_nmf.py:100-169; the
    reference reads the four inputs from pickles - here they are arguments)."""
    aligned_src, aligned_tar = [], []
    for i in range(len(src_W)):
        ia, ib = np.asarray(src_W[i]), np.asarray(tar_W[i])
        if use_stft:
            s, t = np.asarray(src_feat[i]["stft"])[ia], np.asarray(tar_feat[i]["stft"])[ib]
            aligned_src.append({"stft": s, "real": s.real, "imag": s.imag, "fs": src_feat[i].get("fs")})
            aligned_tar.append({"stft": t, "real": t.real, "imag": t.imag, "fs": tar_feat[i].get("fs")})
        else:
            aligned_src.append({k: np.asarray(src_feat[i][k])[ia] for k in ("sp", "ap", "f0")} | {"fs": src_feat[i].get("fs")})
            aligned_tar.append({k: np.asarray(tar_feat[i][k])[ib] for k in ("sp", "ap", "f0")} | {"fs": tar_feat[i].get("fs")})
    return aligned_src, aligned_tar
