"""Dictionary construction of the reference on the GPU: DTW alignment and the gather of aligned frames.

Mirrors /root/reference/01_make_dict_parallel.py:215-249 (`_dtw_alignment`, `dtw_alignment`), :291-292
(`make_exemplar_dict_W`) and /root/reference/04_align_n_nmf.py:100-169 (`align_sp_ap_f0`).  The reference
calls the third-party `dtw` package; its algorithm is restated (see csrc/evc_dtw.hip) - parity with the
package itself is unpinned because it is not installable here.
"""
from __future__ import annotations

import numpy as np

from ..solver import dtw_align, dtw_dictionary, prepare_dictionary


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


def aligned_dictionary(dtw_src, dtw_tar, src_feat, tar_feat, *, use_stft=True, key=None, device=None):
    """`dtw_alignment` + `make_exemplar_dict_W` + `align_sp_ap_f0` + the stacking of `factorize()` / `convert()`
    (01_make_dict_parallel.py:215-249,291-292; 04_align_n_nmf.py:100-169,230-246,320-324,350-361) in one pass on the GPU:
    the DTW paths are consumed on the device and the aligned frames never visit the host (the
    functions above reproduce the reference's step-by-step interfaces through host lists).

      dtw_src[i], dtw_tar[i] : (order, n_frames) alignment features as the reference holds them
      src_feat[i], tar_feat[i]: the scripts' feature dicts; use_stft=True takes |real(f["stft"])| (:320-324), else
                                f[key] for key in "sp" | "ap" | "f0"
    Returns a PreparedDictionary (A = aligned source frames, B = aligned target frames) for `solve_activations` /
    `convert`, and the first dictionary row of every file."""
    if use_stft:
        sa = [np.asarray(f["stft"]) for f in src_feat]
        sb = [np.asarray(f["stft"]) for f in tar_feat]
        A, B, rows = dtw_dictionary([np.asarray(a).T for a in dtw_src], [np.asarray(b).T for b in dtw_tar], sa, sb,
                                    op="abs", real_part=True, device=device)
    else:
        if key not in ("sp", "ap", "f0"):
            raise ValueError("key must be 'sp', 'ap' or 'f0' when use_stft=False")
        col = (lambda v: v[:, np.newaxis] if v.ndim == 1 else v)
        A, B, rows = dtw_dictionary([np.asarray(a).T for a in dtw_src], [np.asarray(b).T for b in dtw_tar],
                                    [col(np.asarray(f[key])) for f in src_feat], [col(np.asarray(f[key])) for f in tar_feat],
                                    device=device)
    return prepare_dictionary(A, B, layout="frame_major", device=device), rows
